"""compute_Sv -> compute_MVBS in ONE pass over the raw power samples (12 B/sample in fp64 instead of
12 + 8 for the two separate calls with echo_range left lazy, 20 + 16 with the array) -- the path BASELINE.json's
metric is quoted on.

``compute_Sv_MVBS`` takes the union of the two reference signatures (calibrate/api.py:249-345 and
commongrid/api.py:30-191) and returns ``(ds_Sv, ds_MVBS)`` exactly as the two calls would;
``ds_Sv["echo_range"]`` is lazy (an ``xr_lite.LazyDeviceArray``: written by ``epa_range_power`` when somebody
reads it; ``materialize_echo_range=True`` runs the two calls instead).  The fused kernel serves power-sample sonars
(EK60, EK80 CW power, AZFP) with the default binning flags; everything else falls back to the two
calls as well -- same results either way.
"""
import numpy as np
import torch

from . import _lib, ops
from .calibrate.api import CALIBRATOR, _compute_cal, _finalize_cal_ds
from .clean.api import remove_background_noise
from .clean.utils import add_remove_background_noise_attrs, extract_dB
from .commongrid.api import _assemble_mvbs, compute_MVBS
from .commongrid.utils import _parse_x_bin, resample_edges
from .utils.prov import echopype_prov_attrs, insert_processing_level
from .xr_lite import DataArray, Dataset, DeferredDataset, DeviceArray, defer_mvbs_enabled, xarray_io


@xarray_io()
def compute_Sv_MVBS(echodata, *, range_bin="20m", ping_time_bin="20s", skipna=True, fill_value=np.nan,
                    closed="left", range_var_max=None, materialize_echo_range=False, env_params=None,
                    cal_params=None, ecs_file=None, waveform_mode=None, encode_mode=None, dtype="float64",
                    device=None, _shard=None, _file=None):
    cal_kw = dict(env_params=env_params, cal_params=cal_params, ecs_file=ecs_file, waveform_mode=waveform_mode,
                  encode_mode=encode_mode, dtype=dtype, device=device)
    mv_kw = dict(range_bin=range_bin, ping_time_bin=ping_time_bin, skipna=skipna, fill_value=fill_value,
                 closed=closed, range_var_max=range_var_max, _shard=_shard)
    is_power = echodata.sonar_model in ("EK60", "ES70", "AZFP") or (
        echodata.sonar_model in ("EK80", "ES80", "EA640") and encode_mode == "power")
    fast = is_power and not materialize_echo_range and skipna and closed == "left" and \
        echodata.sonar_model != "AZFP"  # AZFP rows carry no guard/mask flags -> generic kernel, two calls
    if not fast:  # (EK80 complex samples, AZFP, other binning flags: the two calls -- on a ping shard with the file's scalars)
        ds_Sv = _compute_cal("Sv", echodata, _file=_file, **cal_kw)
        return ds_Sv, compute_MVBS(ds_Sv, **mv_kw)

    # argument checks in the reference's order (_compute_cal, then _setup_and_validate)
    if echodata.sonar_model in ("EK80", "ES80", "EA640") and (waveform_mode is None or encode_mode is None):
        raise ValueError("waveform_mode and encode_mode must be specified for EK80 calibration")
    if not isinstance(range_bin, str):
        raise TypeError("range_bin must be a string")
    range_bin_m = _parse_x_bin(range_bin, "range_bin")
    if not isinstance(ping_time_bin, str):
        raise TypeError("ping_time_bin must be a string")

    cal = CALIBRATOR[echodata.sonar_model](echodata, env_params=env_params, cal_params=cal_params,
                                           ecs_file=ecs_file, waveform_mode=waveform_mode,
                                           encode_mode=encode_mode, dtype=dtype, device=device, file_scalars=_file)
    cal._check_echodata_backscatter_size()
    raw, coef, flags, tau_eff = cal._power_inputs("Sv")
    C, P, S = raw.shape
    ping_time = np.asarray(cal.beam["ping_time"].values).astype("datetime64[ns]", copy=False)
    ns, sorted_valid, _ = ops.ping_time_facts(ping_time, want_device=False)
    if ns.size and not sorted_valid:  # unsorted, or NaT (INT64_MIN)
        if _shard is not None:
            raise NotImplementedError("a ping shard needs sorted, valid ping times")
        ds_Sv = _compute_cal("Sv", echodata, **cal_kw)  # unsorted / NaT pings: generic path
        return ds_Sv, compute_MVBS(ds_Sv, **mv_kw)
    e0, dt, n_t = resample_edges(ping_time, ping_time_bin, sorted_valid=True)
    first_bin = 0

    # range grid np.arange(0, nanmax(echo_range) + bin, bin) (api.py:108-115): run on a conservative
    # grid (largest range any row can reach), get nanmax(echo_range) back as a by-product, trim
    if range_var_max is not None:
        r_cap = _parse_x_bin(range_var_max) + 1e-8
    else:
        r_cap = cal._host_reach_bound(S)  # from host copies of sample_interval / sound_speed: no wait for the GPU
        if r_cap is None:  # parameters in HBM only: reduced on the device, one scalar comes back
            reach = (S - 1) * coef[..., _lib.CF_RA] * coef[..., _lib.CF_RB] + coef[..., _lib.CF_R0]
            reach = torch.nan_to_num(reach, nan=float("-inf"))
            r_cap = float(reach.max().item())
            r_cap = r_cap if r_cap > float("-inf") else float("nan")
    if _shard is not None:
        # the time grid of the whole dataset (this shard covers global bins first_bin .. last_bin) and its range cap:
        # ONE control message
        e0, _, first_bin, last_bin, g_cap = _shard.grid(ns, dt, "left", r_cap if range_var_max is None else float("nan"),
                                                        sorted_valid=True)  # (checked above)
        e0, n_t = e0 + first_bin * dt, last_bin - first_bin + 1
        if range_var_max is None:
            r_cap = g_cap
    bin_start = ops.time_bin_offsets(ops.ping_time_facts(ping_time)[2], e0, dt, n_t)
    n_cap = len(np.arange(0, r_cap + range_bin_m, range_bin_m)) - 1 if np.isfinite(r_cap) else 0
    # degenerate grid (one sample per ping, no valid range) or a kernel that declines (e.g. a range grid too fine for the
    # LDS accumulators): the two calls deal with it.  On a shard the fallback changes the collectives that follow, so
    # the decision is taken ONCE, together, before any rank returns: every rank takes it if any rank must.
    res = None
    if n_cap >= 1:
        try:
            res = ops.sv_mvbs_fused(raw, coef, bin_start, n_t, range_bin_m, n_cap, cal_flags=flags, skipna=True,
                                    closed="left", fill_value=fill_value, dtype=cal.dtype, want_range_max=True,
                                    want_partials=_shard is not None)
        except _lib.EpaError:
            pass
    # (on a shard the plan of the cut-bin exchange and the vote on the fallback travel in ONE control message)
    done = None if _shard is None else _shard.finish(res, first_bin, last_bin, fill_value, shape=(C, n_t, n_cap),
                                                     device=raw.device)
    if (res is None) if _shard is None else done is None:
        ds_Sv = _compute_cal("Sv", echodata, _file=_file, **cal_kw)
        return ds_Sv, compute_MVBS(ds_Sv, **mv_kw)
    # nanmax(echo_range) stays in HBM: on a shard it is all-reduced (MAX) there, behind the kernel; the host reads it
    # when the MVBS dataset is first used (DeferredDataset) -- nothing in this call waits for the GPU
    rmax_t = res["range_max"]
    if _shard is not None and range_var_max is None:
        rmax_t = _shard.range_max_device(rmax_t)
    rmax_f = ops.fetch_async(rmax_t)  # (on its way to the host behind this kernel / all-reduce, on a side stream)
    if _shard is not None:  # bins cut by a shard edge: totals over all ranks, reported by the lowest holder
        res["MVBS"], lo = done
        e0, n_t = e0 + lo * dt, res["MVBS"].shape[1]

    dims = ("channel", "ping_time", "range_sample")
    ds_Sv = Dataset(coords={k: cal.beam.coords[k] for k in dims})
    ds_Sv["Sv"] = DataArray(DeviceArray(res["Sv"]), dims)
    # echo_range is not written: the variable is lazy (coefficient rows + the raw samples' NaN pattern; epa_range_power
    # on first read), as after compute_Sv
    ds_Sv.attrs["echo_range_form"] = "echo_range[c,p,s] = s * sample_interval[c,p] * sound_speed[c,p] / 2 (NaN where Sv input was NaN)"
    ds_Sv["sample_interval"] = cal.beam["sample_interval"]
    if tau_eff is not None:
        ds_Sv["tau_effective"] = DataArray(np.asarray(tau_eff), ("channel",))
    ds_Sv["frequency_nominal"] = cal.beam["frequency_nominal"]
    ds_Sv = cal._add_params_to_output(ds_Sv)
    ds_Sv["echo_range"] = DataArray(cal._lazy_power_range(raw, coef, flags), dims)
    ds_Sv = _finalize_cal_ds(ds_Sv, "Sv", echodata, waveform_mode, encode_mode)

    ds_snap, mv_full = ds_Sv.copy(), res["MVBS"]  # (what the assembly needs, as it is NOW; sums / counts are let go)
    del res

    def build():
        rmax = r_cap if range_var_max is not None else float(rmax_f.item())
        if not np.isfinite(rmax):  # no valid echo_range at all (on any rank): the reference's grid does not exist
            raise ValueError("range bins are empty: the range variable holds no valid values")
        r_edges = np.arange(0, rmax + range_bin_m, range_bin_m)
        n_r = len(r_edges) - 1
        if n_r > n_cap:  # (it must not happen: r_cap bounds every row's reach from above)
            if _shard is not None:
                raise RuntimeError(f"nanmax(echo_range) = {rmax} lies beyond the range grid the shards agreed on "
                                   f"({n_cap} bins of {range_bin_m} m)")
            return compute_MVBS(ds_snap, **{**mv_kw, "_shard": None})  # binned again from the Sv array, exactly
        mvbs_t = mv_full[..., :n_r].contiguous() if n_r != n_cap else mv_full
        return _assemble_mvbs(ds_snap, mvbs_t, "channel", ping_time, e0, dt, n_t, r_edges, "echo_range", range_bin_m,
                              ping_time_bin, "left")

    return ds_Sv, (DeferredDataset(build) if defer_mvbs_enabled() else build())


@xarray_io()
def compute_Sv_clean_MVBS(echodata, ping_num, range_sample_num, *, background_noise_max=None,
                          SNR_threshold="3.0dB", range_bin="20m", ping_time_bin="20s", skipna=True,
                          fill_value=np.nan, closed="left", range_var_max=None, keep_Sv_noise=True,
                          materialize_echo_range=False, env_params=None, cal_params=None, ecs_file=None,
                          waveform_mode=None, encode_mode=None, dtype="float64", device=None):
    """The whole north-star chain ``compute_Sv -> remove_background_noise -> compute_MVBS`` (MVBS of the
    noise-corrected ``Sv_corrected``) in TWO passes over the raw power instead of four array sweeps:

        pass 1  raw -> Sv written once, noise estimate (clean/api.py:397-422) from the values in registers
        pass 2  raw -> Sv_noise / Sv_corrected (api.py:425-430,485-487) written once, MVBS of Sv_corrected
                binned in the same sweep (commongrid/utils.py:592-627)

    24-40 B/sample in fp64 instead of 84.  Returns ``(ds_Sv, ds_MVBS)``: ``ds_Sv`` as
    ``remove_background_noise(compute_Sv(echodata), ...)`` would leave it (``Sv``, ``Sv_corrected``, ``Sv_noise``
    unless ``keep_Sv_noise=False``; ``echo_range`` lazy unless ``materialize_echo_range=True``), ``ds_MVBS`` as
    ``compute_MVBS`` of that dataset with ``Sv := Sv_corrected``.  Served for power-sample EK data with sorted
    pings; anything else runs the three separate calls -- same results either way."""
    cal_kw = dict(env_params=env_params, cal_params=cal_params, ecs_file=ecs_file, waveform_mode=waveform_mode,
                  encode_mode=encode_mode, dtype=dtype, device=device)
    mv_kw = dict(range_bin=range_bin, ping_time_bin=ping_time_bin, skipna=skipna, fill_value=fill_value,
                 closed=closed, range_var_max=range_var_max)

    def separate_calls():
        ds = _compute_cal("Sv", echodata, **cal_kw)
        remove_background_noise(ds, ping_num, range_sample_num, background_noise_max=background_noise_max,
                                SNR_threshold=SNR_threshold)
        corrected = ds.copy()
        corrected["Sv"] = ds["Sv_corrected"]
        return ds, compute_MVBS(corrected, **mv_kw)

    is_power = echodata.sonar_model in ("EK60", "ES70") or (
        echodata.sonar_model in ("EK80", "ES80", "EA640") and encode_mode == "power")
    if not is_power:
        return separate_calls()
    # argument checks in the reference's order
    if echodata.sonar_model in ("EK80", "ES80", "EA640") and (waveform_mode is None or encode_mode is None):
        raise ValueError("waveform_mode and encode_mode must be specified for EK80 calibration")
    nmax = extract_dB(background_noise_max) if background_noise_max is not None else None
    snr = extract_dB(SNR_threshold) if SNR_threshold is not None else None
    if not isinstance(range_bin, str):
        raise TypeError("range_bin must be a string")
    range_bin_m = _parse_x_bin(range_bin, "range_bin")
    if not isinstance(ping_time_bin, str):
        raise TypeError("ping_time_bin must be a string")
    if closed not in ["right", "left"]:
        raise ValueError(f"{closed} is not a valid option. Options are 'left' or 'right'.")

    cal = CALIBRATOR[echodata.sonar_model](echodata, env_params=env_params, cal_params=cal_params,
                                           ecs_file=ecs_file, waveform_mode=waveform_mode,
                                           encode_mode=encode_mode, dtype=dtype, device=device)
    cal._check_echodata_backscatter_size()
    raw, coef, flags, tau_eff = cal._power_inputs("Sv")
    ping_time = np.asarray(cal.beam["ping_time"].values).astype("datetime64[ns]")
    ns = ping_time.astype(np.int64)
    if np.any(np.diff(ns) < 0) or np.isnat(ping_time).any():
        return separate_calls()
    alpha2 = coef[..., _lib.CF_ALPHA2].contiguous()  # 2 * sound_absorption per (channel, ping)

    sv_t, _, noise, rmax = ops.sv_noise_fused(
        raw, coef, alpha2, ping_num, range_sample_num, flags=flags, dtype=cal.dtype,
        noise_max=float("nan") if nmax is None else float(nmax), want_range_max=True)
    e0, dt, n_t = resample_edges(ping_time, ping_time_bin)
    bin_start = ops.time_bin_offsets(ops.to_device(ns), e0, dt, n_t, closed=closed)
    if range_var_max is not None:
        rmax = _parse_x_bin(range_var_max) + 1e-8
    if not np.isfinite(rmax):
        return separate_calls()
    r_edges = np.arange(0, rmax + range_bin_m, range_bin_m)
    n_r = len(r_edges) - 1
    if n_r < 1:  # degenerate grid: the separate calls return the empty MVBS the reference would
        return separate_calls()
    try:
        res = ops.sv_denoise_mvbs(raw, coef, alpha2, noise, ping_num, float(snr), bin_start, n_t, range_bin_m,
                                  n_r, flags=flags, dtype=cal.dtype, skipna=skipna, closed=closed,
                                  fill_value=fill_value, want_noise=keep_Sv_noise,
                                  want_range=materialize_echo_range, want_minmax=True)
    except _lib.EpaError:  # e.g. a range grid too fine for the LDS accumulators
        return separate_calls()

    dims = ("channel", "ping_time", "range_sample")
    ds_Sv = Dataset(coords={k: cal.beam.coords[k] for k in dims})
    ds_Sv["Sv"] = DataArray(DeviceArray(sv_t), dims)
    have_range = materialize_echo_range
    ds_Sv["echo_range"] = DataArray(DeviceArray(res["echo_range"]) if have_range else
                                    cal._lazy_power_range(raw, coef, flags), dims)
    if not have_range:
        ds_Sv.attrs["echo_range_form"] = ("echo_range[c,p,s] = s * sample_interval[c,p] * sound_speed[c,p] / 2 "
                                          "(NaN where Sv input was NaN)")
    ds_Sv["sample_interval"] = cal.beam["sample_interval"]
    if tau_eff is not None:
        ds_Sv["tau_effective"] = DataArray(np.asarray(tau_eff), ("channel",))
    ds_Sv["frequency_nominal"] = cal.beam["frequency_nominal"]
    ds_Sv = cal._add_params_to_output(ds_Sv)
    ds_Sv = _finalize_cal_ds(ds_Sv, "Sv", echodata, waveform_mode, encode_mode)
    mm = res["minmax"]  # actual_range of both outputs comes out of the kernel: no extra sweep
    outs = [("Sv_corrected", res["Sv_corrected"], "corrected", mm[2:4])]
    if keep_Sv_noise:
        outs.insert(0, ("Sv_noise", res["Sv_noise"], "noise", mm[0:2]))
    for name, t, kind, rng_mm in outs:  # clean/api.py:490-502
        ds_Sv[name] = add_remove_background_noise_attrs(DataArray(DeviceArray(t), dims), kind, ping_num,
                                                        range_sample_num, snr, nmax, rng_mm)
    prov = echopype_prov_attrs(process_type="processing")
    prov["processing_function"] = "clean.remove_background_noise"
    ds_Sv.attrs.update(prov)
    ds_Sv = insert_processing_level(ds_Sv, "L*B", input_ds=ds_Sv)
    ds_MVBS = _assemble_mvbs(ds_Sv, res["MVBS"], "channel", ping_time, e0, dt, n_t, r_edges, "echo_range",
                             range_bin_m, ping_time_bin, closed)
    return ds_Sv, ds_MVBS
