"""Helpers of the background-noise functions (mirrors clean/utils.py:13-26, 380-401)."""
import re
from ..xr_lite import LazyAttrs


def extract_dB(dB_str):
    """'3.0dB' -> 3.0 with the reference's error types and messages."""
    if not isinstance(dB_str, str):
        raise TypeError("Decibal input must be a string formatted as `NUMdB` or `NUMdb."
                        f"Cannot be of type `{type(dB_str)}`.")
    match = re.search(r"^[-+]?\d+\.?\d*(?:dB|db)$", dB_str, flags=re.IGNORECASE)
    if match:
        return float(match.group(0)[:-2])
    raise ValueError("Decibal string must be formatted as 'NUMdB' or `NUMdb")


def add_remove_background_noise_attrs(da, sv_type, ping_num, range_sample_num, SNR_threshold, noise_max,
                                      actual_range):
    """clean/utils.py:33-57 of the reference.  ``actual_range``: the (min, max) pair, or a callable returning it -- the
    numbers are by-products of the kernel that writes the array; with a callable the attribute is filled in when
    somebody reads it (``xr_lite.LazyAttrs``) and the call does not wait for the GPU."""
    rounded = lambda mm: [round(float(mm[0]), 2), round(float(mm[1]), 2)]  # noqa: E731
    attrs = LazyAttrs({"long_name": f"Volume backscattering strength, {sv_type} (Sv re 1 m-1)", "units": "dB"})
    if callable(actual_range):
        attrs.set_lazy("actual_range", lambda: rounded(actual_range()))
    else:
        attrs["actual_range"] = rounded(actual_range)
    attrs.update({"noise_ping_num": ping_num, "noise_range_sample_num": range_sample_num,
                  "SNR_threshold": SNR_threshold, "noise_max": noise_max})
    da.attrs = attrs if callable(actual_range) else dict(attrs)
    return da
