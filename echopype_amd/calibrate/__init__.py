from .api import CALIBRATOR, compute_Sv, compute_TS  # noqa: F401

__all__ = ["compute_Sv", "compute_TS", "CALIBRATOR"]
