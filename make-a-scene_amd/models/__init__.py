from .vqvae import VQBASE
