"""Helpers of the background-noise functions (mirrors clean/utils.py:13-26, 380-401)."""
import re


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
    da.attrs = {
        "long_name": f"Volume backscattering strength, {sv_type} (Sv re 1 m-1)",
        "units": "dB",
        "actual_range": [round(float(actual_range[0]), 2), round(float(actual_range[1]), 2)],
        "noise_ping_num": ping_num,
        "noise_range_sample_num": range_sample_num,
        "SNR_threshold": SNR_threshold,
        "noise_max": noise_max,
    }
    return da
