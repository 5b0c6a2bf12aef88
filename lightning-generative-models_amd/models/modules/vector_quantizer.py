"""Path-compatible home of the vector quantisers (reference: models/modules/vector_quantizer.py).
The HIP-backed implementations live next to the VQ-VAE that drives them."""
from models.generative.vae.vqvae import VectorQuantizer, VectorQuantizerEMA  # noqa: F401
