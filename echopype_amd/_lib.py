"""ctypes binding of libechopype_amd.so (the C ABI declared in include/echopype_amd.h).

The HIP library is the product: if it is missing, or fails to load, importing this module
raises -- there is NO CPU fallback anywhere in echopype_amd.

torch is imported first on purpose: PyTorch-ROCm bundles its own libamdhip64.so.7; loading it
before our library makes the dynamic linker resolve our DT_NEEDED libamdhip64.so.7 to the copy
already in the process, so torch tensors (device memory, streams, RCCL) and our kernels share one
HIP runtime.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# ECHOPYPE_AMD_LIB: alternative build of the same library (A/B tuning of kernel variants)
LIB_PATH = os.environ.get("ECHOPYPE_AMD_LIB") or os.path.join(_HERE, "lib", "libechopype_amd.so")

EPA_OK, EPA_EINVAL, EPA_EHIP, EPA_ENOMEM, EPA_EUNSUPPORTED = range(5)
F32, F64 = 0, 1
CAL_SV, CAL_TS = 0, 1
SONAR_EK60, SONAR_EK80 = 0, 1
PM_SCALAR, PM_CHANNEL, PM_CHANNEL_PING, PM_PULSE_TABLE = range(4)
NCOEF = 8
CF_RA, CF_RB, CF_R0, CF_SHIFT, CF_ALPHA2, CF_A0, CF_G, CF_D = range(8)
NCCOEF = 8
CC_RA, CC_RB, CC_SHIFT, CC_ALPHA2, CC_A, CC_PSCALE = range(6)
FLAG_GUARD_POS, FLAG_MASK_RANGE = 1, 2
BIN_SKIPNA, BIN_CLOSED_RIGHT, BIN_RANGE_AS_STORED = 1, 2, 4
POOL_NANMEAN, POOL_NANMEDIAN = 0, 1
CCP = ("sample_interval", "tau_nominal", "transmit_power", "sound_speed", "absorption", "gain", "freq_center", "psi",
       "sa_correction", "z_er", "z_et", "angle_offset_alongship", "angle_offset_athwartship", "beamwidth_alongship",
       "beamwidth_athwartship")  # enum epa_ccoef_param
EK80_NFFT = 2048
APPLY_MASKS_WS_DOUBLES = 131072  # EPA_APPLY_MASKS_WS_DOUBLES


class EpaError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python echopype_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  echopype_amd has no CPU fallback."
        )
    return ctypes.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _u, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_double, ctypes.c_size_t
_i64 = ctypes.c_int64

# name -> argtypes; every function returns int status except epa_version / epa_last_error
SIGNATURES = {
    "epa_version": [],
    "epa_source_digest": [],
    "epa_last_error": [],
    "epa_launch_trace": [_i],
    "epa_last_range_stats_filled": [],
    "epa_launch_seen": [],
    "epa_device_count": [ctypes.POINTER(_i)],
    "epa_set_device": [_i],
    "epa_device_name": [_i, ctypes.c_char_p, _sz],
    "epa_malloc": [ctypes.POINTER(_vp), _sz],
    "epa_free": [_vp],
    "epa_memset": [_vp, _i, _sz, _vp],
    "epa_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "epa_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "epa_stream_synchronize": [_vp],
    "epa_timer_create": [ctypes.POINTER(_vp)],
    "epa_timer_destroy": [_vp],
    "epa_timer_start": [_vp, _vp],
    "epa_timer_stop": [_vp, _vp],
    "epa_timer_elapsed_ms": [_vp, ctypes.POINTER(ctypes.c_float)],
    "epa_power_coef_ek": [_i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp,
                          _vp, _i, _vp, _i, _i, _vp, _vp],
    "epa_pulse_table_lookup": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "epa_sv_power": [_vp, _vp, _i, _i, _i, _i, _u, _vp, _vp, _i, _vp],
    "epa_sv_power_stats": [_vp, _vp, _i, _i, _i, _i, _u, _vp, _vp, _i, _vp, _vp, _vp],
    "epa_range_power": [_vp, _vp, _i, _i, _i, _u, _vp, _i, _vp],
    "epa_range_complex": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp],
    "epa_sv_complex_cw_stats": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "epa_time_bin_offsets": [_vp, _i, _i64, _i64, _i, _u, _vp, _vp],
    "epa_sv_mvbs_fused": [_vp, _vp, _i, _i, _i, _i, _u, _vp, _vp, _i, _d, _i, _u, _d, _vp, _vp, _vp,
                          _vp, _vp, _vp, _vp, _i, _vp],
    "epa_sv_mvbs_fused_i16": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _i, _d, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "epa_sv_mvbs_fused_depth": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                _vp],
    "epa_mvbs": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _d, _i, _u, _d, _vp, _vp, _vp, _i, _vp],
    "epa_selftest_lin_from_db": [_vp, _vp, _sz, _vp],
    "epa_selftest_log10": [_vp, _vp, _sz, _vp],
    "epa_selftest_log10_inline": [_vp, _vp, _sz, _vp],
    "epa_mvbs_finalize": [_vp, _vp, _sz, _d, _vp, _i, _vp],
    "epa_edge_pack": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "epa_edge_prepare_max": [_vp, _i, _vp],
    "epa_edge_gather": [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "epa_edge_finalize_mvbs": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _i, _vp],
    "epa_affine_rows": [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "epa_depth_rows": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "epa_nanminmax": [_vp, _sz, _i, _vp, _vp, _vp],
    "epa_mvbs_index": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "epa_noise_estimate": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _vp],
    "epa_noise_finalize": [_vp, _vp, _i, _i, _d, _vp, _vp],
    "epa_noise_apply": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _vp],
    "epa_noise_apply_rows": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _vp],
    "epa_complex_coef_ek80": [_i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp],
    "epa_sv_complex": [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "epa_sv_complex_indexed": [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "epa_sv_complex_fft": [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "epa_sv_complex_fft_indexed": [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i,
                                   _vp, _vp, _vp],
    "epa_range_bin_smooth": [_vp, _vp, _i, _i, _i, _i, _d, _d, _i, _vp, _i, _vp],
    "epa_impulse_mask": [_vp, _i, _i, _i, _i, _d, _vp, _i, _vp],
    "epa_pool_sv": [_vp, _i, _i, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _i, _vp],
    "epa_attenuated_mask": [_vp, _vp, _i, _i, _i, _d, _d, _i, _d, _vp, _i, _vp],
    "epa_apply_mask": [_vp, _vp, _sz, _sz, _d, _vp, _sz, _vp, _i, _vp],
    "epa_apply_masks": [_vp, _vp, _vp, _i, _sz, _d, _vp, _sz, _vp, _vp, _vp, _i, _vp],
    "epa_mask_and": [_vp, _vp, _sz, _sz, _vp, _vp],
    "epa_range_step_mean": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "epa_first_not_le": [_vp, _sz, _d, _i, _vp, _vp],
    "epa_range_rows_check": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "epa_sv_noise_fused": [_vp, _vp, _vp, _i, _i, _i, _i, _u, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "epa_denoise_mvbs": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _vp, _vp, _i, _d, _i, _u, _d, _vp, _vp, _vp,
                         _vp, _vp, _i, _vp],
    "epa_sv_denoise_mvbs": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _u, _i, _i, _d, _vp, _vp, _i, _d, _i, _u, _d, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp, _i, _vp],
    "epa_geodesic_steps": [_vp, _vp, _i, _vp, _vp],
    "epa_nasc": [_vp, _vp, _i, _i, _i, _vp, _i, _d, _i, _u, _vp, _vp, _vp, _vp, _i, _vp],
    "epa_pool_sv_value": [_vp, _vp, _vp, _i, _i, _i, _d, _i, _d, _d, _d, _i, _d, _vp, _vp, _vp, _i, _vp],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the .so does not export a declared symbol
    _fn.argtypes = _args
    _fn.restype = ctypes.c_char_p if _name in ("epa_last_error", "epa_launch_trace", "epa_launch_seen", "epa_source_digest") else _i


def _check_source_digest():
    """The library must have been built from the sources lying beside it (csrc/, include/): a stale binary raises here
    instead of answering for code it was not made from.  Skipped for an alternative build handed in through
    ECHOPYPE_AMD_LIB (A/B tuning: other flags on purpose) and when the sources are not there (an installed copy)."""
    if os.environ.get("ECHOPYPE_AMD_LIB"):
        return
    import importlib.util

    spec = importlib.util.spec_from_file_location("_epa_build", os.path.join(_HERE, "build.py"))
    if spec is None or not os.path.isdir(os.path.join(_HERE, "csrc")):
        return
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    want, got = b.source_digest(), lib.epa_source_digest().decode()
    if want != got:
        raise ImportError(f"{LIB_PATH} was built from other sources (digest {got}, the tree has {want}): "
                          "rebuild it with `python echopype_amd/build.py` (`--force` recompiles every translation unit)")


_check_source_digest()


class launch_trace:
    """``with launch_trace() as t: ...; t.kernels`` -- the names of the kernels the calls inside launched (tests)."""

    def __enter__(self):
        lib.epa_launch_trace(1)
        self.kernels = []
        return self

    def __exit__(self, *exc):
        self.kernels = [k for k in lib.epa_launch_trace(2).decode().split(";") if k]
        lib.epa_launch_trace(0)
        return False


def launched_kernels():
    """Names of every kernel this process has launched through the library so far."""
    return [k for k in lib.epa_launch_seen().decode().split(";") if k]


def last_error() -> str:
    msg = lib.epa_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int, what: str = "") -> None:
    """Map a non-zero C status to a Python exception (SURVEY 8b error conventions)."""
    if status == EPA_OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if status == EPA_EINVAL:
        raise ValueError(msg)
    if status == EPA_ENOMEM:
        raise MemoryError(msg)
    raise EpaError(f"[status {status}] {msg}")


def call(name: str, *args) -> None:
    check(getattr(lib, name)(*args), name)
