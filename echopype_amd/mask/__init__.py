from .api import apply_mask  # noqa: F401

__all__ = ["apply_mask"]
