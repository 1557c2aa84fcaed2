"""Host-side mirror of the one function of the reference's ``datasets`` package the explicit-triplet path needs."""
from .articles import similar_articles  # noqa: F401
