"""Public calibration entry points with the reference's signatures
(/root/reference/echopype/calibrate/api.py: CALIBRATOR :11-18, _compute_cal :23-246, compute_Sv
:249-345, compute_TS :348-449).  Two extra keyword-only arguments: ``dtype`` ("float64" |
"float32", the arithmetic and output type) and ``device`` (torch device; default current GPU).
Outputs stay resident in HBM (``DeviceArray``) until ``.values`` is read.
"""
import logging

import numpy as np

from ..utils.prov import echopype_prov_attrs
from .calibrate_azfp import CalibrateAZFP
from .calibrate_ek import CalibrateEK60, CalibrateEK80

from ..xr_lite import xarray_io  # noqa: E402

logger = logging.getLogger("echopype_amd.calibrate")

CALIBRATOR = {"EK60": CalibrateEK60, "EK80": CalibrateEK80, "AZFP": CalibrateAZFP, "ES70": CalibrateEK60,
              "ES80": CalibrateEK80, "EA640": CalibrateEK80}


def check_input_args_combination(waveform_mode, encode_mode, pulse_compression=None):
    """echodata/simrad.py:12-51."""
    if waveform_mode not in ["CW", "BB"]:
        raise ValueError("The input waveform_mode must be either 'CW' or 'BB'!")
    if encode_mode not in ["complex", "power"]:
        raise ValueError("The input encode_mode must be either 'complex' or 'power'!")
    if waveform_mode == "BB" and encode_mode == "power":
        raise ValueError("Data from broadband ('BB') transmission must be recorded as complex samples")
    if pulse_compression is not None:
        if pulse_compression and (waveform_mode != "BB" or encode_mode != "complex"):
            raise RuntimeError("Pulse compression can only be used with "
                               "waveform_mode='BB' and encode_mode='complex'")


def _compute_cal(cal_type, echodata, env_params=None, cal_params=None, ecs_file=None, waveform_mode=None,
                 encode_mode=None, assume_single_filter_time=None, drop_last_hanning_zero=False,
                 dtype="float64", device=None, fft_dtype=None, _file=None):
    """(``_file``: the whole-file facts of a ping shard, echopype_amd.sharding.file_scalars -- set by sharding.compute_Sv.)"""
    waveform_mode = "BB" if waveform_mode == "FM" else waveform_mode
    if echodata.sonar_model == "EK80":
        if waveform_mode is None or encode_mode is None:
            raise ValueError("waveform_mode and encode_mode must be specified for EK80 calibration")
        check_input_args_combination(waveform_mode=waveform_mode, encode_mode=encode_mode)
    elif echodata.sonar_model in ("EK60", "AZFP"):
        if waveform_mode is not None and waveform_mode != "CW":
            logger.warning("This sonar model transmits only narrowband signals (waveform_mode='CW'). "
                           "Calibration will be in CW mode")
        if encode_mode is not None and encode_mode != "power":
            logger.warning("This sonar model only record data as power or power/angle samples "
                           "(encode_mode='power'). Calibration will be done on the power samples.")
    if (echodata.sonar_model != "EK80" or encode_mode != "complex") and assume_single_filter_time is not None:
        raise ValueError("assume_single_filter_time can only be used on complex EK80 data.")
    if echodata.sonar_model not in CALIBRATOR:
        raise ValueError(f"unsupported sonar_model {echodata.sonar_model!r}")

    def _compute_cal_ds(slice_dict):
        cal_obj = CALIBRATOR[echodata.sonar_model](
            echodata, env_params=env_params, cal_params=cal_params, ecs_file=ecs_file,
            waveform_mode=waveform_mode, encode_mode=encode_mode, slice_dict=slice_dict,
            drop_last_hanning_zero=drop_last_hanning_zero, dtype=dtype, device=device, fft_dtype=fft_dtype,
            file_scalars=_file)
        cal_obj._check_echodata_backscatter_size()
        return cal_obj.compute_Sv() if cal_type == "Sv" else cal_obj.compute_TS()

    vend = echodata["Vendor_specific"]
    n_filter = vend.sizes.get("filter_time", 1) if echodata.sonar_model in ("EK80", "ES80", "EA640") else 1
    if n_filter <= 1:
        cal_ds = _compute_cal_ds({})
    else:
        from .calibrate_ek import retrieve_correct_beam_group

        beam = echodata[retrieve_correct_beam_group(echodata, waveform_mode, encode_mode)]
        tau = np.asarray(beam["transmit_duration_nominal"].values, dtype=np.float64)
        pt = np.asarray(beam["ping_time"].values).astype("datetime64[ns]")
        chans = list(beam["channel"].values)
        if assume_single_filter_time:
            # filter set of each channel's first valid ping (api.py:101-123) -- of the WHOLE file on a ping shard
            if _file is not None and _file.get("first_valid_ping_time") is not None:
                first = {ch: np.datetime64(int(_file["first_valid_ping_time"][i]), "ns") for i, ch in enumerate(chans)}
            else:
                first = {ch: pt[np.flatnonzero(~np.isnan(tau[i]))[0]] for i, ch in enumerate(chans)}
            cal_ds = _compute_cal_ds({"first_valid_filter_time_per_channel": first})
        else:
            # every (channel, filter interval) pair in ONE pass over the whole grid (the reference calibrates them one by
            # one and merges, api.py:125-197): per-pair replicas / effective pulse lengths, a replica index per ping
            cal_ds = _compute_cal_ds({"filter_intervals": True})

    return _finalize_cal_ds(cal_ds, cal_type, echodata, waveform_mode, encode_mode)


def _finalize_cal_ds(cal_ds, cal_type, echodata, waveform_mode, encode_mode):
    """Attributes, provenance and water_level of a calibrated dataset (api.py:199-244)."""
    # attributes (api.py:199-219)
    cal_ds.coords["range_sample"].attrs = {"long_name": "Along-range sample number, base 0"}
    cal_ds.data_vars["echo_range"].attrs = {"long_name": "Range distance", "units": "m"}
    cal_ds.data_vars[cal_type].attrs = {
        "long_name": {"Sv": "Volume backscattering strength (Sv re 1 m-1)",
                      "TS": "Target strength (TS re 1 m^2)"}[cal_type],
        "units": "dB",
    }
    if echodata.sonar_model == "EK80":
        cal_ds.data_vars[cal_type].attrs.update({"waveform_mode": waveform_mode, "encode_mode": encode_mode})
    # provenance (api.py:221-240)
    if echodata.source_file is not None:
        source_file = echodata.source_file
    elif echodata.converted_raw_path is not None:
        source_file = echodata.converted_raw_path
    else:
        source_file = "SOURCE FILE NOT IDENTIFIED"
    prov = echopype_prov_attrs(process_type="processing")
    prov["processing_function"] = f"calibrate.compute_{cal_type}"
    files = [source_file] if isinstance(source_file, str) else list(source_file)
    cal_ds["source_filenames"] = (("filenames",), np.asarray([str(f) for f in files]),
                                  {"long_name": "Source filenames"})
    cal_ds = cal_ds.assign_attrs(prov)
    if "water_level" in echodata["Platform"].data_vars:
        cal_ds["water_level"] = echodata["Platform"]["water_level"]
    return cal_ds


@xarray_io()
def compute_Sv(echodata, *, env_params=None, cal_params=None, ecs_file=None, waveform_mode=None, encode_mode=None,
               assume_single_filter_time=None, drop_last_hanning_zero=False, dtype="float64", device=None,
               fft_dtype=None):
    """Volume backscattering strength Sv.  The reference's ``compute_Sv(echodata, **kwargs)`` forwards to
    ``_compute_cal`` (calibrate/api.py:23-33, 249-345); the keywords it accepts there are spelled out here, keyword-only
    as they effectively are -- ``inspect.signature`` and tab completion show them, a misspelt one is a ``TypeError`` at
    the call (tests/test_signatures.py compares with the reference's source).  Three more for the accelerated path:
    ``dtype`` ("float64" | "float32": arithmetic and output type), ``device`` (torch device, default the current GPU)
    and, EK80 broadband, ``fft_dtype`` = arithmetic of the pulse-compression transform ("float64" | "float32", default
    = dtype -- a complex64 transform is as precise as a float32 output, with errors relative to the strongest echo of
    each 2048-sample tile; a float64 output always gets complex128 unless asked otherwise)."""
    return _compute_cal("Sv", echodata, env_params=env_params, cal_params=cal_params, ecs_file=ecs_file,
                        waveform_mode=waveform_mode, encode_mode=encode_mode,
                        assume_single_filter_time=assume_single_filter_time,
                        drop_last_hanning_zero=drop_last_hanning_zero, dtype=dtype, device=device, fft_dtype=fft_dtype)


@xarray_io()
def compute_TS(echodata, *, env_params=None, cal_params=None, ecs_file=None, waveform_mode=None, encode_mode=None,
               assume_single_filter_time=None, drop_last_hanning_zero=False, dtype="float64", device=None,
               fft_dtype=None):
    """Target strength TS (calibrate/api.py:348-449); keywords as for :func:`compute_Sv`."""
    return _compute_cal("TS", echodata, env_params=env_params, cal_params=cal_params, ecs_file=ecs_file,
                        waveform_mode=waveform_mode, encode_mode=encode_mode,
                        assume_single_filter_time=assume_single_filter_time,
                        drop_last_hanning_zero=drop_last_hanning_zero, dtype=dtype, device=device, fft_dtype=fft_dtype)
