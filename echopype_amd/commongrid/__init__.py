from .api import compute_MVBS, compute_MVBS_index_binning, compute_NASC  # noqa: F401

__all__ = ["compute_MVBS", "compute_MVBS_index_binning", "compute_NASC"]
