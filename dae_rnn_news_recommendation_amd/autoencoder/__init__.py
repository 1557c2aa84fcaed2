"""Mirror of the reference's ``autoencoder`` package (estimators + the helper modules they use)."""
from .autoencoder import DenoisingAutoencoder
from .autoencoder_triplet import DenoisingAutoencoderTriplet  # noqa: F401
