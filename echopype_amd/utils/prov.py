"""Provenance attributes stamped on hot-path outputs (reference: utils/prov.py:24-43 and the
processing-level decorator :181-331; only the attrs are reproduced, not the subsystem)."""
import datetime as _dt

SOFTWARE_NAME = "echopype_amd"
SOFTWARE_VERSION = "0.1.0"


def echopype_prov_attrs(process_type="processing"):
    return {
        f"{process_type}_software_name": SOFTWARE_NAME,
        f"{process_type}_software_version": SOFTWARE_VERSION,
        f"{process_type}_time": _dt.datetime.now(_dt.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ"),
    }


def _has_valid_position(ds):
    import numpy as np

    if "latitude" not in ds or "longitude" not in ds:
        return False
    return bool(np.isfinite(ds["latitude"].values).any() and np.isfinite(ds["longitude"].values).any())


def insert_processing_level(ds, level, input_ds=None):
    """``processing_level`` is only set when the data carry valid positions (prov.py:181-308);
    '*' in the level inherits the digit/letter of the input (e.g. L3* from Level 2A -> Level 3A)."""
    src = input_ds if input_ds is not None else ds
    if not _has_valid_position(src):
        return ds
    in_level = src.attrs.get("processing_level")
    if "*" in level:
        if not in_level:
            return ds
        suffix = in_level[-1]
        if level == "L*B":
            new = f"Level {in_level.split()[-1][0]}B"
        else:
            new = f"Level {level[1]}{suffix}"
    else:
        new = f"Level {level[1:]}"
    ds.attrs["processing_level"] = new
    ds.attrs["processing_level_url"] = "https://echopype.readthedocs.io/en/stable/processing-levels.html"
    return ds
