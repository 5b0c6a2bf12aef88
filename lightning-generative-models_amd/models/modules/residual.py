"""Path-compatible home of the residual stack (reference: models/modules/residual.py)."""
from models.generative.vae.vqvae import ResidualBlock, ResidualStack  # noqa: F401
