from .api import add_depth, swap_dims_channel_frequency  # noqa: F401

__all__ = ["add_depth", "swap_dims_channel_frequency"]
