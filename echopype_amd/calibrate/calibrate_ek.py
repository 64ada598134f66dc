"""EK60 / EK80 calibrators: host-side parameter assembly, GPU sample passes.

Mirrors /root/reference/echopype/calibrate/calibrate_ek.py (CalibrateEK :56-206, CalibrateEK60
:209-265, CalibrateEK80 :268-710).  The constructors do what the reference's do (beam-group
selection, env then cal params) -- that is O(C*P) host work.  The O(C*P*S) part of
``_cal_power_samples`` / ``_cal_complex_samples`` is ONE kernel launch through the C ABI:
epa_power_coef_ek + epa_sv_power, or epa_sv_complex (matched filter + power + Sv/TS).
"""
import logging

import numpy as np
import torch

from .. import _lib, ops
from ..echodata import BEAM1
from ..xr_lite import DataArray, Dataset, DeviceArray, LazyDeviceArray, host_readable
from .cal_params import PulseTableParam, get_cal_params_EK
from .calibrate_base import ECHO_DIMS, CalibrateBase, cp_array
from .ek80_complex import get_filter_coeff, get_tau_effective, get_transmit_signal
from .env_params import get_env_params_EK

logger = logging.getLogger("echopype_amd.calibrate")


def retrieve_correct_beam_group(echodata, waveform_mode, encode_mode):
    """Beam group for the requested modes, with the reference's checks (echodata/simrad.py:54-179)."""
    if echodata.sonar_model in ("EK60", "ES70"):
        if waveform_mode != "CW":
            raise RuntimeError("Incorrect waveform_mode input provided!")
        if encode_mode != "power":
            raise RuntimeError("Incorrect encode_mode input provided!")
        if "backscatter_i" in echodata[BEAM1].variables:
            raise RuntimeError("Provided echodata object does not correspond to an EK60-like "
                               "sensor, but is labeled as data from an EK60-like sensor!")
        return BEAM1
    if echodata.sonar_model in ("EK80", "ES80", "EA640"):
        if "waveform_encode_descr" not in echodata["Sonar"]:
            raise ValueError("Echodata missing `waveform_encode_descr`. "
                             "Reconvert using the latest Echopype version.")
        descr = np.asarray(echodata["Sonar"]["waveform_encode_descr"].values).astype(str)
        match = "power" if encode_mode == "power" else ("complex_CW" if waveform_mode == "CW" else "complex_FM")
        idx = np.flatnonzero(descr == match)
        if idx.size == 0:
            raise RuntimeError(f"No beam group with the specified encode_mode {encode_mode} "
                               f"and waveform_mode {waveform_mode} found in the provided echodata!")
        return f"Sonar/Beam_group{int(idx[0]) + 1}"
    raise RuntimeError("EchoData was produced by a non-Simrad or unknown Simrad echo sounder!")


def _index_of(ds, dim, label):
    vals = np.asarray(ds[dim].values)
    hit = np.flatnonzero(vals == (np.asarray(label).astype(vals.dtype) if vals.dtype.kind == "M" else label))
    if hit.size == 0:
        raise KeyError(f"{label!r} not found along {dim!r}")
    return int(hit[0])


def _slice_beam_vend(beam, vend, slice_dict):
    """One channel, one filter interval (calibrate_ek.py:25-35): pings in [start, end] inclusive."""
    ci = _index_of(beam, "channel", slice_dict["channel"])
    pt = np.asarray(beam["ping_time"].values).astype("datetime64[ns]")
    start, end = slice_dict["beam_group_start_time"], slice_dict["beam_group_end_time"]
    keep = pt >= np.datetime64(start, "ns")
    if end is not None and not (isinstance(end, np.datetime64) and np.isnat(end)):
        keep &= pt <= np.datetime64(end, "ns")
    idx = np.flatnonzero(keep)
    beam = beam.isel(channel=[ci], ping_time=idx)
    vend = vend.isel(filter_time=_index_of(vend, "filter_time", slice_dict["filter_time"]))
    return beam, vend


def _collapse_vend(vend, slice_dict):
    """assume_single_filter_time: per channel keep the filter set valid at its first valid ping
    (calibrate_ek.py:38-53): the filter_time dimension collapses."""
    from ..xr_lite import Dataset as _DS

    chans = list(vend["channel"].values)
    picks = [_index_of(vend, "filter_time", slice_dict["first_valid_filter_time_per_channel"][ch]) for ch in chans]
    out = _DS(coords={k: c for k, c in vend.coords.items() if k != "filter_time"})
    for name, da in vend.data_vars.items():
        if "filter_time" not in da.dims:
            out.data_vars[name] = da
            continue
        a = np.asarray(da.values)
        cax, fax = da.dims.index("channel"), da.dims.index("filter_time")
        a = np.moveaxis(a, (cax, fax), (0, 1))
        sel = np.stack([a[i, picks[i]] for i in range(len(chans))], axis=0)
        dims = ("channel",) + tuple(d for d in da.dims if d not in ("channel", "filter_time"))
        out[name] = (dims, sel, dict(da.attrs))
    return out


class CalibrateEK(CalibrateBase):
    def __init__(self, echodata, env_params, cal_params, ecs_file, **kwargs):
        super().__init__(echodata, env_params, cal_params, ecs_file, **kwargs)
        self.ed_beam_group = None
        self.slice_dict = {}
        self.beam = None
        self.vend = None
        self._range_dev = None

    # -- shared per-(channel, ping) inputs ---------------------------------------------------------
    def _shape(self):
        sz = self.beam["backscatter_r"].shape
        return sz[0], sz[1], sz[2]

    def _cp(self, v, name):
        C, P, _ = self._shape()
        return cp_array(v, C, P, name)

    def _gpt_mask(self):
        C = self._shape()[0]
        if self.sonar_type == "EK60":
            return np.ones(C, dtype=bool)
        if "transceiver_type" in self.vend:
            return np.asarray(self.vend["transceiver_type"].values).astype(str) == "GPT"
        return np.zeros(C, dtype=bool)

    def compute_echo_range(self):
        """echo_range (range.py:98-157) is produced by the same kernel pass as Sv/TS; it is
        materialised here only if asked for before calibration."""
        self.range_meter = None

    def _tau_effective(self, flag_complex):
        """Effective pulse length per channel (calibrate_ek.py:113-151 / :583-607)."""
        tdn = self.beam["transmit_duration_nominal"]
        if isinstance(tdn.data, DeviceArray) and not host_readable(tdn.data) and tuple(tdn.dims) == ("channel", "ping_time"):
            tau_nom0 = tdn.data.tensor[:, 0].double().cpu().numpy()  # C values
        else:
            tau_nom0 = np.asarray(self._cp(tdn, "transmit_duration_nominal"))[:, 0]
        fsc = self.file_scalars or {}
        if fsc.get("tau_nominal_first_ping") is not None:  # a ping shard: ping 0 of the WHOLE file
            tau_nom0 = np.asarray(fsc["tau_nominal_first_ping"], dtype=np.float64).reshape(tau_nom0.shape)
        try:
            coeff = get_filter_coeff(self.vend)
            fs = self.cal_params["receiver_sampling_frequency"]  # KeyError for EK60 -> fallback
            tx, tx_time = get_transmit_signal(self.beam, coeff, self.waveform_mode, fs,
                                              getattr(self, "drop_last_hanning_zero", False),
                                              whole_file=fsc.get("transmit_params"))
            te = get_tau_effective(tx, {k: 1 / np.diff(v[:2]) for k, v in tx_time.items()},
                                   self.waveform_mode, self.beam["channel"]).values.copy()
        except Exception as e:  # noqa: BLE001 - same catch-all as the reference
            mode = "complex" if flag_complex else "power"
            logger.warning("Could not compute tau_effective from transmit signal in %s encoding mode; "
                           "falling back to transmit_duration_nominal. Error: %s", mode, repr(e))
            te, tx = tau_nom0.copy(), None
        gpt = self._gpt_mask()
        te[gpt] = tau_nom0[gpt]
        return te, tx

    def _finish(self, cal_type, out_t, range_t, tau_eff, range_stats=None):
        C, P, S = self._shape()
        ds = Dataset(coords={k: self.beam.coords[k] for k in ECHO_DIMS})
        ds[cal_type] = self._wrap(out_t, ECHO_DIMS)
        # nanmin / nanmax / NaN count of echo_range travel with the array (what compute_MVBS asks for next)
        ds["echo_range"] = self._wrap(range_t, ECHO_DIMS, stats=range_stats)
        self.range_meter = ds["echo_range"]
        if cal_type == "Sv":
            tau_eff, te_dims = np.asarray(tau_eff, dtype=np.float64), ("channel",)
            if tau_eff.ndim == 2:
                # several filter intervals: one value per channel when all its intervals agree -- what the reference's
                # merge of the slices needs (xr.merge(compat="no_conflicts"), calibrate/api.py:190-194) -- else the grid
                ok = ~np.isnan(tau_eff)
                first = np.array([row[m][0] if m.any() else np.nan for row, m in zip(tau_eff, ok)])
                same = np.all((tau_eff == first[:, None]) | ~ok)
                pairs = getattr(self, "_pair_te", None)
                if pairs:  # (every pair of the file, also those whose pings another shard holds: the file's answer)
                    for ci in range(len(first)):
                        vals = [v for c_, v in pairs if c_ == ci and not np.isnan(v)]
                        if vals:
                            first[ci] = vals[0]
                            same = same and all(v == vals[0] for v in vals)
                if same:
                    tau_eff = first
                else:
                    te_dims = ("channel", "ping_time")
            ds["tau_effective"] = DataArray(tau_eff, te_dims, attrs=dict(
                long_name="Effective pulse length", units="s",
                description="Effective pulse length used for Sv. GPT uses transmit_duration_nominal."))
        ds["frequency_nominal"] = self.beam["frequency_nominal"]
        return self._add_params_to_output(ds)

    def _power_inputs(self, cal_type):
        """(raw f32 tensor, coefficient rows, kernel flags, tau_eff): everything the power-sample
        kernels need; shared by compute_Sv/TS and the fused Sv->MVBS entry point."""
        C, P, S = self._shape()
        f64 = torch.float64
        tau_eff, _ = self._tau_effective(False) if cal_type == "Sv" else (np.ones(C), None)
        gpt = self._gpt_mask()
        cpd = lambda v, name: self._cp_dev(v, C, P, name, f64)  # noqa: E731
        # equivalent_beam_angle: per channel as in the files, or (channel, ping_time) -- the reference broadcasts any
        # cal parameter of that shape into CSv (calibrate_ek.py:154-162); K0 takes either (EPA_PM_CHANNEL[_PING])
        psi_da = self.cal_params["equivalent_beam_angle"]
        if getattr(psi_da, "ndim", 0) > 1:
            psi_t = cpd(psi_da, "equivalent_beam_angle")
        else:
            psi_t = self._dev(np.asarray(psi_da.values, dtype=np.float64).reshape(-1), f64)
        # gain / sa_correction straight from the Vendor_specific pulse-length tables: looked up per ping by the kernel
        g, sa = self.cal_params["gain_correction"], self.cal_params["sa_correction"]
        tables = (isinstance(g, PulseTableParam) and isinstance(sa, PulseTableParam) and not g.materialized
                  and not sa.materialized and np.array_equal(g.pulse_length, sa.pulse_length, equal_nan=True))
        if tables:
            kw = dict(pulse_length=self._dev(g.pulse_length, f64), gain_is_table=True, sa_is_table=True)
            g_t, sa_t = self._dev(g.table, f64), self._dev(sa.table, f64)
        else:
            kw = {}
            g_t, sa_t = cpd(g, "gain_correction"), cpd(sa, "sa_correction")
        si_t = cpd(self.beam["sample_interval"], "sample_interval")
        plan = getattr(self, "_plan", None)
        if plan is not None and (plan["replica_id"] < 0).any():  # pings no filter interval covers: NaN rows
            si_h = cp_array(self.beam["sample_interval"], C, P).copy()
            si_h[plan["replica_id"] < 0] = np.nan
            si_t = self._dev(si_h, f64)
        coef = ops.power_coef_ek(
            si_t,
            cpd(self.beam["transmit_duration_nominal"], "transmit_duration_nominal"),
            cpd(self.beam["transmit_power"], "transmit_power"),
            cpd(self.env_params["sound_speed"], "sound_speed"),
            cpd(self.env_params["sound_absorption"], "sound_absorption"),
            g_t, sa_t, psi_t, self._dev(np.asarray(self.beam["frequency_nominal"].values, float), f64),
            self._dev(np.asarray(tau_eff, float), f64),
            sonar=self.sonar_type, cal_type=cal_type,
            gpt=self._dev(gpt.astype(np.uint8)) if self.sonar_type == "EK80" else None, **kw)
        raw = self._dev(self.beam["backscatter_r"].data, torch.float32)
        return raw, coef, _lib.FLAG_GUARD_POS | _lib.FLAG_MASK_RANGE, tau_eff

    def _cal_power_samples(self, cal_type):
        """One fused pass for calibrate_ek.py:79-206."""
        raw, coef, flags, tau_eff = self._power_inputs(cal_type)
        # (float64 only: the fused kernel bins on the range in double, compute_MVBS on a float32 dataset on the range
        # rounded to float32 as the array would hold it -- a sample on a bin edge could change sides)
        if cal_type == "Sv" and self.dtype == torch.float64 and self.defer_enabled() and \
                ops.sv_power_vectorized(raw, raw.shape[2], self.dtype):
            out_t, range_t = self._deferred_sv_power(raw, coef, cal_type, flags)
            return self._finish(cal_type, out_t, range_t, tau_eff)
        out_t, range_t, stats = self._sv_power_lazy_range(raw, coef, cal_type, flags)
        return self._finish(cal_type, out_t, range_t, tau_eff, range_stats=stats)


class CalibrateEK60(CalibrateEK):
    def __init__(self, echodata, env_params, cal_params, ecs_file, **kwargs):
        super().__init__(echodata, env_params, cal_params, ecs_file, **kwargs)
        self.sonar_type = "EK60"
        self.waveform_mode = "CW"
        self.encode_mode = "power"
        self.ed_beam_group = retrieve_correct_beam_group(self.echodata, self.waveform_mode, self.encode_mode)
        self.beam = self.echodata[self.ed_beam_group]
        self.vend = self.echodata["Vendor_specific"]
        self.env_params = get_env_params_EK("EK60", self.beam, self.echodata["Environment"], self.env_params)
        self.cal_params = get_cal_params_EK("CW", self.beam["frequency_nominal"], self.beam, self.vend,
                                            self.cal_params, sonar_type="EK60")
        self.compute_echo_range()

    def compute_Sv(self, **kwargs):
        return self._cal_power_samples("Sv")

    def compute_TS(self, **kwargs):
        return self._cal_power_samples("TS")


class CalibrateEK80(CalibrateEK):
    EK80_params = {"z_et": 75, "z_er": 1000}

    def __init__(self, echodata, env_params, cal_params, waveform_mode, encode_mode, ecs_file=None,
                 slice_dict=None, drop_last_hanning_zero=False, **kwargs):
        super().__init__(echodata, env_params, cal_params, ecs_file, **kwargs)
        self.sonar_type = "EK80"
        self.waveform_mode = waveform_mode
        self.encode_mode = encode_mode
        self.slice_dict = slice_dict or {}
        self.drop_last_hanning_zero = drop_last_hanning_zero
        self.ed_beam_group = retrieve_correct_beam_group(self.echodata, waveform_mode, encode_mode)
        self.beam = self.echodata[self.ed_beam_group]
        self.vend = self.echodata["Vendor_specific"]
        # multi-filter_time files (calibrate_ek.py:323-333): one (channel, filter interval) slice,
        # or the filter set of each channel's first valid ping when a single set is assumed
        self._plan = None
        if "channel" in self.slice_dict:
            self.beam, self.vend = _slice_beam_vend(self.beam, self.vend, self.slice_dict)
        if self.slice_dict.get("filter_intervals"):
            # every (channel, filter interval) pair of the file in ONE pass over the whole (channel, ping_time) grid:
            # the reference calibrates slice by slice and merges (calibrate/api.py:125-197); here each pair gets its own
            # replica / effective pulse length and every ping the index of its pair (_interval_plan, _tau_effective)
            self.vend_full = self.vend
        if "first_valid_filter_time_per_channel" in self.slice_dict:
            self.vend = _collapse_vend(self.vend, self.slice_dict)
        elif "filter_time" in self.vend.sizes:
            self.vend = self.vend.isel(filter_time=0)
        bch = list(map(str, self.beam["channel"].values))
        vch = list(map(str, self.vend["channel"].values))
        if bch != vch:  # vend.sel(channel=beam.channel) (:333)
            self.vend = self.vend.isel(channel=[vch.index(c) for c in bch])
            if self.slice_dict.get("filter_intervals"):
                self.vend_full = self.vend_full.isel(channel=[vch.index(c) for c in bch])
        if self.slice_dict.get("filter_intervals"):
            self._plan = self._interval_plan()
        C, P = self.beam["backscatter_r"].shape[:2]
        self._fc_collapsed = False
        if waveform_mode == "BB":
            f0 = cp_array(self.beam["transmit_frequency_start"], C, P)
            f1 = cp_array(self.beam["transmit_frequency_stop"], C, P)
            tdn = self.beam["transmit_duration_nominal"]
            tdn_host = host_readable(tdn.data)
            tau = cp_array(tdn, C, P) if tdn_host else None
            const = lambda a: a is not None and bool(np.all((a == a[:, :1]) | np.isnan(a)))  # noqa: E731
            # The usual file sweeps the same band with the same pulse on every ping: the (channel, ping_time) centre
            # frequency of calibrate_ek.py:336-340 is then one number per channel, and everything derived from it
            # (absorption formula, calibration tables interpolated at it) is evaluated per channel -- same values,
            # broadcast back to (channel, ping_time) where the output dataset carries them (_add_params_to_output).
            if P > 1 and const(f0) and const(f1) and const(tau):
                self.freq_center = DataArray(((f0 + f1) / 2)[:, 0].copy(), ("channel",))
                self._fc_collapsed = True
            else:
                self.freq_center = DataArray((f0 + f1) / 2, ("channel", "ping_time"))  # :336-340
        else:
            self.freq_center = self.beam["frequency_nominal"]
        self.env_params = get_env_params_EK("EK80", self.beam, self.echodata["Environment"], self.env_params,
                                            freq=self.freq_center)
        self.cal_params = get_cal_params_EK(waveform_mode, self.freq_center, self.beam, self.vend,
                                            self.cal_params, sonar_type="EK80")
        self.compute_echo_range()

    def _shape(self):
        sz = self.beam["backscatter_r"].shape
        return sz[0], sz[1], sz[2]

    # -- a file with several filter_time entries ----------------------------------------------------------------------
    def _interval_plan(self):
        """The (channel, filter interval) pairs the reference loops over (calibrate/api.py:133-160): per channel the
        filter_time stamps that are ping times with a valid transmit_duration_nominal start an interval that runs to
        1 ns before the next one (the last to the end of the file).  Returns dict(pairs=[(channel index, filter_time
        index, ping indices)], replica_id (C, P) int32 -- the pair of every ping, -1 for a ping no interval covers)."""
        C, P = self.beam["backscatter_r"].shape[:2]
        tau = cp_array(self.beam["transmit_duration_nominal"], C, P)
        pt = np.asarray(self.beam["ping_time"].values).astype("datetime64[ns]")
        ft_all = np.asarray(self.vend_full["filter_time"].values).astype("datetime64[ns]")
        ft_sorted = np.sort(ft_all)
        pairs, rid = [], np.full((C, P), -1, dtype=np.int32)
        fsc = self.file_scalars or {}
        for ci in range(C):
            if fsc.get("interval_starts") is not None:  # a ping shard: the stamps that start an interval ANYWHERE in the file
                starts = np.sort(ft_all[np.asarray(fsc["interval_starts"][ci], dtype=bool)])
            else:
                starts = np.intersect1d(pt[~np.isnan(tau[ci])], ft_sorted)
            for k, start in enumerate(starts):
                keep = pt >= start
                if k + 1 < len(starts):
                    keep &= pt <= starts[k + 1] - np.timedelta64(1, "ns")
                idx = np.flatnonzero(keep)
                # (a ping shard may hold no ping of the interval: the pair stays in the plan -- its effective pulse length
                #  is part of the file's answer for the channel, see _finish)
                if idx.size == 0 and fsc.get("interval_starts") is None:
                    continue
                rid[ci, idx] = len(pairs)
                pairs.append((ci, int(np.flatnonzero(ft_all == start)[0]), idx))
        return dict(pairs=pairs, replica_id=rid)

    def _pair_views(self, ci, fi, idx):
        """(beam parameters, vend) of one (channel, filter interval) pair: the few host arrays get_transmit_signal and
        get_filter_coeff read -- never the samples."""
        C, P = self.beam["backscatter_r"].shape[:2]
        ch = np.asarray(self.beam["channel"].values)[ci:ci + 1]
        pt = np.asarray(self.beam["ping_time"].values)[idx]
        small = Dataset(coords={"channel": ch, "ping_time": pt})
        for name in ("transmit_duration_nominal", "slope", "transmit_frequency_start", "transmit_frequency_stop",
                     "transmit_type"):
            if name in self.beam:
                a = np.asarray(self.beam[name].values)
                small[name] = (("channel", "ping_time"), a[ci:ci + 1][:, idx])
        small["frequency_nominal"] = (("channel",), np.asarray(self.beam["frequency_nominal"].values)[ci:ci + 1])
        return small, self.vend_full.isel(filter_time=fi, channel=[ci])

    def _tau_effective(self, flag_complex):
        if self._plan is None:
            return super()._tau_effective(flag_complex)
        # per (channel, interval): the replica of THAT filter set and its effective pulse length, with the reference's
        # fallback to the nominal pulse length of the slice's first ping (calibrate_ek.py:113-151); GPT channels nominal
        C, P = self.beam["backscatter_r"].shape[:2]
        tau = cp_array(self.beam["transmit_duration_nominal"], C, P)
        gpt = self._gpt_mask()
        fs_all = self.cal_params.get("receiver_sampling_frequency")
        te = np.full((C, P), np.nan)
        txs = []
        fsc = self.file_scalars or {}
        self._pair_te = []  # (channel index, effective pulse length) of every pair of the FILE
        for ci, fi, idx in self._plan["pairs"]:
            tau0 = tau[ci, idx[0]] if idx.size else np.nan
            whole = None
            if fsc.get("interval_tau0") is not None:  # a ping shard: the INTERVAL's first ping, its transmit parameters
                tau0 = float(fsc["interval_tau0"][ci, fi])
                whole = {k: (v[0][ci:ci + 1, fi], v[1][ci:ci + 1, fi]) for k, v in fsc["interval_transmit_params"].items()}
            tx = None
            try:
                beam_k, vend_k = self._pair_views(ci, fi, idx)
                fs = np.asarray(getattr(fs_all, "values", fs_all), dtype=np.float64)
                fs = fs if fs.ndim == 0 else fs[ci:ci + 1]
                tx_d, tx_time = get_transmit_signal(beam_k, get_filter_coeff(vend_k), self.waveform_mode, fs,
                                                    self.drop_last_hanning_zero, whole_file=whole)
                val = get_tau_effective(tx_d, {k: 1 / np.diff(v[:2]) for k, v in tx_time.items()}, self.waveform_mode,
                                        beam_k["channel"]).values[0]
                tx = np.asarray(next(iter(tx_d.values())))
            except Exception as e:  # noqa: BLE001 - same catch-all as the reference
                mode = "complex" if flag_complex else "power"
                logger.warning("Could not compute tau_effective from transmit signal in %s encoding mode; "
                               "falling back to transmit_duration_nominal. Error: %s", mode, repr(e))
                val = tau0
            te[ci, idx] = tau0 if gpt[ci] else val
            self._pair_te.append((ci, tau0 if gpt[ci] else val))
            txs.append(tx)
        return te, txs

    _FC_KEYS = ("sound_absorption", "gain_correction", "equivalent_beam_angle", "impedance_transducer",
                "angle_offset_alongship", "angle_offset_athwartship", "angle_sensitivity_alongship",
                "angle_sensitivity_athwartship", "beamwidth_alongship", "beamwidth_athwartship")

    def _add_params_to_output(self, ds_out):
        """Parameters evaluated per channel because the centre frequency does not change along ping_time go out with
        the (channel, ping_time) dimensions the reference gives them (zero-copy broadcast views)."""
        ds_out = self._add_params(ds_out)
        if self._plan is not None and (self._plan["replica_id"] < 0).any():
            # pings no filter interval covers are in none of the reference's slices: NaN after its outer join
            hole = self._plan["replica_id"] < 0
            for name, da in list(ds_out.data_vars.items()):
                if tuple(da.dims) == ("channel", "ping_time") and da.dtype.kind == "f":
                    a = np.array(da.values, dtype=np.float64)
                    a[hole] = np.nan
                    ds_out[name] = DataArray(a, da.dims, attrs=da.attrs)
        return ds_out

    def _add_params(self, ds_out):
        if not self._fc_collapsed:
            return super()._add_params_to_output(ds_out)
        C, P, _ = self._shape()
        saved = {}
        for group in (self.env_params, self.cal_params):
            for key in self._FC_KEYS:
                v = group.get(key)
                if isinstance(v, DataArray) and tuple(v.dims) == ("channel",) and not isinstance(v.data, DeviceArray):
                    saved[(id(group), key)] = (group, v)
                    group[key] = DataArray(np.broadcast_to(np.asarray(v.values)[:, None], (C, P)), ("channel", "ping_time"),
                                           attrs=v.attrs)
        try:
            return super()._add_params_to_output(ds_out)
        finally:
            for (_, key), (group, v) in saved.items():
                group[key] = v

    def _get_B_theta_phi_m(self):
        """Transceiver gain compensation for BB mode (calibrate_ek.py:507-530)."""
        cp = self.cal_params
        with np.errstate(invalid="ignore", divide="ignore"):
            fa = (np.abs(-self._cp(cp["angle_offset_alongship"], "angle_offset_alongship"))
                  / (self._cp(cp["beamwidth_alongship"], "beamwidth_alongship") / 2)) ** 2
            ft = (np.abs(-self._cp(cp["angle_offset_athwartship"], "angle_offset_athwartship"))
                  / (self._cp(cp["beamwidth_athwartship"], "beamwidth_athwartship") / 2)) ** 2
            B = 0.5 * 6.0206 * (fa + ft - 0.18 * fa * ft)
        return np.where(np.isnan(B), 0.0, B)

    def _complex_inputs(self, cal_type):
        """Everything epa_sv_complex needs, assembled on the host from O(C*P) parameters: the sample planes on the
        device, the (C, P, 6) coefficient rows, the flattened replicas + offsets (BB), tau_effective."""
        C, P, S = self._shape()
        B = self.beam["backscatter_r"].shape[3]
        bb = self.waveform_mode == "BB"
        tau_eff, tx = self._tau_effective(True)
        plan = self._plan
        if plan is not None and bb and any(t is None for t in tx):
            raise ValueError("a filter interval without a usable transmit replica: broadband samples cannot be "
                             "pulse-compressed (assume_single_filter_time=True calibrates with the first filter set)")
        if plan is None and tx is None:
            coeff = get_filter_coeff(self.vend)
            tx, _ = get_transmit_signal(self.beam, coeff, self.waveform_mode,
                                        self.cal_params["receiver_sampling_frequency"],
                                        self.drop_last_hanning_zero,
                                        whole_file=(self.file_scalars or {}).get("transmit_params"))
        # the (C, P, 8) coefficient rows are built on the device (epa_complex_coef_ek80): parameters go up in the shape
        # they have -- scalar, (C,), or (C, P) (straight from HBM when the echodata is resident) -- no (C, P) NumPy math
        def dv(v, name):
            d = v.data if isinstance(v, DataArray) else None
            if isinstance(d, DeviceArray) and tuple(v.dims) == ("channel", "ping_time") and d.shape == (C, P):
                return d.tensor
            a = np.asarray(getattr(v, "values", v), dtype=np.float64)
            dims = tuple(v.dims) if isinstance(v, DataArray) else None
            if a.ndim == 0 or (a.ndim == 1 and a.shape[0] == C and dims != ("ping_time",) and not (P == C and dims is None)):
                return self._dev(np.ascontiguousarray(a), torch.float64)
            return self._dev(cp_array(v, C, P, name), torch.float64)

        cpar, env = self.cal_params, self.env_params
        si_param = dv(self.beam["sample_interval"], "sample_interval")
        if plan is not None and (plan["replica_id"] < 0).any():  # uncovered pings: a NaN row -> NaN Sv and echo_range
            si_h = cp_array(self.beam["sample_interval"], C, P).copy()
            si_h[plan["replica_id"] < 0] = np.nan
            si_param = self._dev(si_h, torch.float64)
        params = dict(sample_interval=si_param,
                      tau_nominal=dv(self.beam["transmit_duration_nominal"], "transmit_duration_nominal"),
                      transmit_power=dv(self.beam["transmit_power"], "transmit_power"),
                      sound_speed=dv(env["sound_speed"], "sound_speed"), absorption=dv(env["sound_absorption"], "sound_absorption"),
                      gain=dv(cpar["gain_correction"], "gain_correction"), freq_center=dv(self.freq_center, "freq_center"),
                      psi=dv(cpar["equivalent_beam_angle"], "equivalent_beam_angle"),
                      sa_correction=None if bb else dv(cpar["sa_correction"], "sa_correction"),
                      z_er=dv(cpar["impedance_transceiver"], "impedance_transceiver"),
                      z_et=dv(cpar["impedance_transducer"], "impedance_transducer"))
        if bb:
            for k in ("angle_offset_alongship", "angle_offset_athwartship", "beamwidth_alongship", "beamwidth_athwartship"):
                params[k] = dv(cpar[k], k)
        cc_t = ops.complex_coef_ek80(params, self._dev(np.asarray(tau_eff, dtype=np.float64), torch.float64), C, P, B=B, bb=bb,
                                     cal_type=cal_type, gpt=self._dev(self._gpt_mask().astype(np.uint8)))
        rep = off = rid = None
        max_taps = 0
        if bb:
            chans = list(self.beam["channel"].values)
            taps = [np.asarray(tx[ch]) for ch in chans] if plan is None else [np.asarray(t) for t in tx]
            if plan is not None:
                rid = self._dev(np.ascontiguousarray(plan["replica_id"]))
            off_h = np.concatenate([[0], np.cumsum([t.size for t in taps])]).astype(np.int32)
            flat = np.concatenate(taps).astype(np.complex64)
            rep = self._dev(np.ascontiguousarray(flat.view(np.float32)))
            off = self._dev(off_h)
            max_taps = int(max(t.size for t in taps))
        re = self._dev(self.beam["backscatter_r"].data)
        im = self._dev(self.beam["backscatter_i"].data)
        if re.dtype not in (torch.float32, torch.float64):
            re, im = re.double(), im.double()
        return dict(re=re, im=im, ccoef=cc_t, replica=rep, replica_off=off, max_taps=max_taps, replica_id=rid), tau_eff

    def _cal_complex_samples(self, cal_type):
        """One fused pass for calibrate_ek.py:532-659 (+ ek80_complex.py:285-391 for BB)."""
        k, tau_eff = self._complex_inputs(cal_type)
        # the LDS-FFT form and the CW kernel leave {nanmin, nanmax, NaN count} of echo_range as a by-product: the array can stay
        # lazy then (written by epa_range_complex if somebody reads it; compute_MVBS bins through the coefficient rows)
        lazy = k["replica"] is None or ops.sv_complex_uses_fft(k["replica"], k["max_taps"])
        res = ops.sv_complex(k["re"], k["im"], k["ccoef"], replica=k["replica"], replica_off=k["replica_off"],
                             max_taps=k["max_taps"], cal_type=cal_type, dtype=self.dtype, fft_dtype=self.fft_dtype,
                             want_range=not lazy, want_range_stats=True, replica_id=k["replica_id"])
        range_t = res["echo_range"]
        if lazy:
            re, ccoef, dtype, version = k["re"], k["ccoef"], self.dtype, k["re"]._version

            def make():
                if re._version != version:
                    raise RuntimeError("echo_range was left lazy by compute_Sv/compute_TS and backscatter_r has been "
                                       "modified in place since: its NaN mask can no longer be reproduced")
                return ops.range_complex(re, ccoef, dtype=dtype)

            range_t = LazyDeviceArray(res["out"].shape, dtype, re.device, make, stats=res["range_stats"],
                                      rows=ops.power_rows_of_complex(ccoef))
            # (range.py:138-160: s * sample_interval * sound_speed / 2 minus a non-negative offset, clipped at 0)
            range_t.reach_bound = self._host_reach_bound(res["out"].shape[2])
        return self._finish(cal_type, res["out"], range_t, tau_eff, range_stats=res["range_stats"])

    def _compute_cal(self, cal_type):
        flag_complex = self.waveform_mode == "BB" or self.encode_mode == "complex"
        return self._cal_complex_samples(cal_type) if flag_complex else self._cal_power_samples(cal_type)

    def compute_Sv(self):
        return self._compute_cal("Sv")

    def compute_TS(self):
        return self._compute_cal("TS")
