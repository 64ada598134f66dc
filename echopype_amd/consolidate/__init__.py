from .api import add_depth  # noqa: F401

__all__ = ["add_depth"]
