"""echopype_amd -- MI355X (gfx950) native implementation of echopype's array-compute hot path

    calibrate.compute_Sv / compute_TS -> clean.remove_background_noise -> commongrid.compute_MVBS

behind echopype's own function signatures.  Host code is Python; the (channel, ping_time,
range_sample) arrays go through a C ABI (include/echopype_amd.h, loaded with ctypes) to hand-written
HIP kernels.  There is no CPU fallback: importing the package loads libechopype_amd.so and fails
loudly if it has not been built (``python echopype_amd/build.py``).
"""
import os as _os

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) when it initialises;
# streams that share a queue run one after the other.  ``pipeline`` puts consecutive files on two side streams next to
# the upload / download streams and the caller's own -- with four queues two of them can end up on ONE queue (measured,
# round 6: the headline's side-by-side launches then serialise, 0.67 instead of 0.72 of the roofline).  Eight queues, unless
# the caller has said otherwise; it has to be in the environment before the first HIP call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402  (loads the HIP library; raises if missing)
from . import calibrate, clean, commongrid, consolidate, mask, ops, pipeline, synth, utils  # noqa: F401,E402
from .echodata import EchoData  # noqa: F401,E402
from .fused import compute_Sv_clean_MVBS, compute_Sv_MVBS  # noqa: F401,E402
from .xr_lite import DataArray, Dataset, DeviceArray  # noqa: F401,E402

__version__ = "0.1.0"
__all__ = ["calibrate", "clean", "commongrid", "consolidate", "mask", "utils", "ops", "synth", "pipeline", "compute_Sv_MVBS", "compute_Sv_clean_MVBS", "EchoData", "Dataset", "DataArray",
           "DeviceArray"]
