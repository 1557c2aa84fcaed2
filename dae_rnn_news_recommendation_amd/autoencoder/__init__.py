"""Mirror of the reference's ``autoencoder`` package (estimators + the helper modules they use)."""
from .autoencoder import DenoisingAutoencoder  # noqa: F401
