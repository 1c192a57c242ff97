"""gan_amd: MI355X-native convolution hot path (forward, data gradient, weight gradient, double backward) for the
GAN training scripts of the reference, behind drop-in torch.nn / torch.nn.functional look-alikes."""
from . import data, functional, nn, ops, optim  # noqa: F401
from .graph import GraphedStep  # noqa: F401
from .ops import get_backward_overlap, get_precision, set_backward_overlap, set_precision, set_split_reduction  # noqa: F401

__all__ = ["data", "functional", "nn", "ops", "optim", "set_precision", "get_precision", "set_backward_overlap", "get_backward_overlap", "set_split_reduction", "GraphedStep"]
