"""A minimal stand-in for ``echopype.echodata.EchoData`` (read API only).

The reference's EchoData is an ``xr.DataTree`` wrapper around SONAR-netCDF4 groups
(/root/reference/echopype/echodata/echodata.py:43-730, out of scope -- SURVEY 2 #17).  The hot path
only *reads* it: ``ed["Sonar/Beam_group1"]``, ``ed["Vendor_specific"]``, ``ed["Environment"]``,
``ed["Platform"]``, ``ed["Sonar"]``, ``.sonar_model``, ``.source_file``, ``.converted_raw_path``
(calibrate/api.py:39,98,226-244).  This class offers exactly that over
:class:`echopype_amd.xr_lite.Dataset` groups; a real ``EchoData`` (when echopype is importable)
can be passed to the calibrators unchanged because they only use this read API.
"""
import numpy as np

from .xr_lite import DataArray, Dataset, from_xarray

BEAM1 = "Sonar/Beam_group1"
BEAM2 = "Sonar/Beam_group2"


class EchoData:
    def __init__(self, sonar_model, groups=None, source_file=None, converted_raw_path=None):
        self.sonar_model = sonar_model
        self.source_file = source_file
        self.converted_raw_path = converted_raw_path
        self._groups = {k: from_xarray(v) for k, v in (groups or {}).items()}
        self._groups.setdefault("Platform", Dataset())
        self._groups.setdefault("Sonar", Dataset())

    def __getitem__(self, group):
        try:
            return self._groups[group]
        except KeyError:
            raise KeyError(f"EchoData has no group {group!r}; groups: {sorted(self._groups)}") from None

    def __setitem__(self, group, ds):
        self._groups[group] = from_xarray(ds)

    def __contains__(self, group):
        return group in self._groups

    @property
    def group_paths(self):
        return list(self._groups)

    def to_device(self, device=None, groups=(BEAM1, BEAM2, "Environment")):
        """Move the floating-point array variables of the beam and Environment groups into HBM (DeviceArray), in place;
        returns self.  The samples AND the per-(channel, ping) parameters (sample_interval, transmit power / duration,
        sound speed, absorption ...) then stay resident between calls: compute_Sv / compute_Sv_MVBS read them where
        they are instead of assembling and uploading (channel, ping_time) arrays on every call (the analogue of keeping
        the reference's dask arrays persisted).  Coordinates and string / integer variables stay on the host."""
        from . import ops
        from .xr_lite import DeviceArray

        for gname in groups:
            if gname not in self._groups:
                continue
            ds = self._groups[gname]
            for name, da in list(ds.data_vars.items()):
                if isinstance(da.data, DeviceArray) or da.ndim < 2 or da.dtype.kind != "f":
                    continue
                host = np.asarray(da.data)
                # per-(channel, ping) parameters keep a read-only view of their host copy: the host logic that only has
                # to LOOK at them (is the pulse the same on every ping?) then never copies anything back
                mirror = None
                if host.nbytes <= 64 * 1024 * 1024:
                    mirror = host.view()
                    mirror.flags.writeable = False
                ds.data_vars[name] = DataArray(DeviceArray(ops.to_device(host, device=device), host=mirror), da.dims,
                                               da.coords, da.attrs, name)
            # the ping_time coordinate: resident like the samples -- its int64 twin goes up now and stays with the (from
            # here on read-only) host array, so that compute_MVBS neither scans nor uploads it per call
            if gname != "Environment" and "ping_time" in ds.coords:
                pt = ds.coords["ping_time"].values
                if isinstance(pt, np.ndarray) and pt.dtype == np.dtype("datetime64[ns]"):
                    pt.flags.writeable = False
                    ops.ping_time_facts(pt)
        return self

    def __repr__(self):
        return f"<EchoData sonar_model={self.sonar_model!r} groups={self.group_paths}>"


KNOWN_GROUPS = ("Top-level", "Environment", "Platform", "Platform/NMEA", "Provenance", "Sonar", BEAM1, BEAM2,
                "Vendor_specific")


def as_lite_echodata(ed):
    """echopype's own EchoData (an xr.DataTree wrapper: groups are xarray Datasets) -> this module's EchoData over
    lite Datasets; the calibrators only use the read API both share."""
    if isinstance(ed, EchoData):
        return ed
    groups = {}
    for g in getattr(ed, "group_paths", KNOWN_GROUPS):
        try:
            ds = ed[g]
        except Exception:  # noqa: BLE001 - a group this file does not have
            continue
        if ds is not None:
            groups[g] = from_xarray(ds)
    return EchoData(ed.sonar_model, groups, source_file=getattr(ed, "source_file", None),
                    converted_raw_path=getattr(ed, "converted_raw_path", None))


def _time1(ping_time):
    return np.asarray(ping_time, dtype="datetime64[ns]")


def from_ek60_arrays(d, source_file="synthetic_ek60.raw"):
    """Build an EK60 EchoData from the arrays of :func:`echopype_amd.synth.ek60_numpy`, with the
    variable names / dims the converter writes (convert/set_groups_ek60.py:88-152,578-667,728-787)."""
    ch = list(d["channel"])
    pt = _time1(d["ping_time"])
    C, P, S = d["backscatter_r"].shape
    beam = Dataset(coords={"channel": ch, "ping_time": pt, "range_sample": np.arange(S)})
    beam["backscatter_r"] = (("channel", "ping_time", "range_sample"), d["backscatter_r"])
    for k in ("sample_interval", "transmit_duration_nominal", "transmit_power"):
        beam[k] = (("channel", "ping_time"), np.asarray(d[k], dtype=np.float64))
    beam["frequency_nominal"] = (("channel",), np.asarray(d["frequency_nominal"], dtype=np.float64))
    beam["equivalent_beam_angle"] = (("channel",), np.asarray(d["equivalent_beam_angle"], dtype=np.float64))
    K = d["pulse_length"].shape[1]
    vend = Dataset(coords={"channel": ch, "pulse_length_bin": np.arange(K)})
    for k in ("pulse_length", "gain_correction", "sa_correction"):
        vend[k] = (("channel", "pulse_length_bin"), np.asarray(d[k], dtype=np.float64))
    env = Dataset(coords={"channel": ch, "time1": pt})
    env["sound_speed_indicative"] = (("channel", "time1"), np.asarray(d["sound_speed_indicative"], dtype=np.float64))
    env["absorption_indicative"] = (("channel", "time1"), np.asarray(d["absorption_indicative"], dtype=np.float64))
    return EchoData("EK60", {BEAM1: beam, "Vendor_specific": vend, "Environment": env},
                    source_file=source_file)


def from_ek80_arrays(d, filters, encode="complex", source_file="synthetic_ek80.raw", filter_time_idx=None):
    """EK80 EchoData (complex samples in Beam_group1; convert/set_groups_ek80.py:796-840,967-1068,
    1234-1518).  ``filters`` = dict(wbt_fil, wbt_decifac, pc_fil, pc_decifac) applied to every channel."""
    ch = list(d["channel"])
    pt = _time1(d["ping_time"])
    C, P, S, B = d["backscatter_r"].shape
    bb = d.get("waveform", "BB") == "BB"
    beam = Dataset(coords={"channel": ch, "ping_time": pt, "range_sample": np.arange(S), "beam": np.arange(B)})
    dims4 = ("channel", "ping_time", "range_sample", "beam")
    beam["backscatter_r"] = (dims4, d["backscatter_r"])
    beam["backscatter_i"] = (dims4, d["backscatter_i"])
    cp = ("channel", "ping_time")
    beam["sample_interval"] = (cp, np.asarray(d["sample_interval"], dtype=np.float64))
    beam["transmit_duration_nominal"] = (cp, np.tile(np.asarray(d["tau"], float)[:, None], (1, P)))
    beam["transmit_power"] = (cp, np.tile(np.asarray(d["transmit_power"], float)[:, None], (1, P)))
    beam["slope"] = (cp, np.tile(np.asarray(d["slope"], float)[:, None], (1, P)))
    f0 = np.asarray(d["f_start"] if bb else d["frequency_nominal"], float)
    f1 = np.asarray(d["f_stop"] if bb else d["frequency_nominal"], float)
    beam["transmit_frequency_start"] = (cp, np.tile(f0[:, None], (1, P)))
    beam["transmit_frequency_stop"] = (cp, np.tile(f1[:, None], (1, P)))
    beam["transmit_type"] = (cp, np.full((C, P), "LFM" if bb else "CW"))
    beam["frequency_nominal"] = (("channel",), np.asarray(d["frequency_nominal"], float))
    beam["equivalent_beam_angle"] = (("channel",), np.asarray(d["psi"], float))
    for k_out, k_in in (("angle_offset_alongship", "angle_offset_alongship"),
                        ("angle_offset_athwartship", "angle_offset_athwartship"),
                        ("beamwidth_twoway_alongship", "beamwidth_alongship"),
                        ("beamwidth_twoway_athwartship", "beamwidth_athwartship")):
        beam[k_out] = (("channel",), np.asarray(d[k_in], float))
    beam["angle_sensitivity_alongship"] = (("channel",), np.full(C, 23.0))
    beam["angle_sensitivity_athwartship"] = (("channel",), np.full(C, 23.0))
    nw, npc = filters["wbt_fil"].size, filters["pc_fil"].size
    vend = Dataset(coords={"channel": ch, "pulse_length_bin": np.arange(5), "WBT_filter_n": np.arange(nw + 3),
                           "PC_filter_n": np.arange(npc + 2)})

    vend_dim = {"WBT_coeffs_real": "WBT_filter_n", "WBT_coeffs_imag": "WBT_filter_n",
                "PC_coeffs_real": "PC_filter_n", "PC_coeffs_imag": "PC_filter_n"}

    def _pad(c, n):  # NaN padded like the converter (set_groups_ek80.py:1411-1518)
        out = np.full((C, n), np.nan)
        out[:, : c.size] = c
        return out

    vend["WBT_coeffs_real"] = (("channel", "WBT_filter_n"), _pad(filters["wbt_fil"].real, nw + 3))
    vend["WBT_coeffs_imag"] = (("channel", "WBT_filter_n"), _pad(filters["wbt_fil"].imag, nw + 3))
    vend["PC_coeffs_real"] = (("channel", "PC_filter_n"), _pad(filters["pc_fil"].real, npc + 2))
    vend["PC_coeffs_imag"] = (("channel", "PC_filter_n"), _pad(filters["pc_fil"].imag, npc + 2))
    vend["WBT_deci_fac"] = (("channel",), np.full(C, filters["wbt_decifac"]))
    vend["PC_deci_fac"] = (("channel",), np.full(C, filters["pc_decifac"]))
    if filter_time_idx is not None:
        # the same filter set recorded at several filter_time stamps (a file whose filter datagrams
        # were re-sent): (channel, filter_time, n) as in set_groups_ek80.py:1411-1518
        ft = pt[np.asarray(filter_time_idx)]
        vend.coords["filter_time"] = DataArray(ft, ("filter_time",), name="filter_time")
        for k in ("WBT_coeffs_real", "WBT_coeffs_imag", "PC_coeffs_real", "PC_coeffs_imag"):
            a = vend.data_vars[k].values
            vend.data_vars.pop(k)
            vend[k] = (("channel", "filter_time", vend_dim[k]), np.repeat(a[:, None, :], len(ft), axis=1))
        for k in ("WBT_deci_fac", "PC_deci_fac"):
            a = vend.data_vars[k].values
            vend.data_vars.pop(k)
            vend[k] = (("channel", "filter_time"), np.repeat(a[:, None], len(ft), axis=1))
    vend["transceiver_type"] = (("channel",), np.array(d.get("transceiver_type", ["WBT"] * C)))
    vend["impedance_transceiver"] = (("channel",), np.asarray(d["z_er"], float))
    vend["receiver_sampling_frequency"] = (("channel",), np.asarray(d["fs"], float))
    pl = np.tile(np.array([256e-6, 512e-6, 1024e-6, 2048e-6, 4096e-6]), (C, 1))
    g = np.asarray(d["gain"], float)
    vend["pulse_length"] = (("channel", "pulse_length_bin"), pl)
    vend["gain_correction"] = (("channel", "pulse_length_bin"), np.stack([g - 1, g - .5, g, g + .2, g + .3], axis=1))
    vend["sa_correction"] = (("channel", "pulse_length_bin"), np.tile(np.asarray(d["sa"], float)[:, None], (1, 5)))
    env = Dataset(coords={"time1": pt[:1]})
    env["sound_speed_indicative"] = (("time1",), np.array([float(np.asarray(d["sound_speed"]).flat[0])]))
    env["temperature"] = (("time1",), np.array([10.0]))
    env["salinity"] = (("time1",), np.array([35.0]))
    env["depth"] = (("time1",), np.array([10.0]))
    env["acidity"] = (("time1",), np.array([8.0]))
    sonar = Dataset(coords={"beam_group": ["Beam_group1"]})
    descr = "power" if encode == "power" else ("complex_FM" if bb else "complex_CW")
    sonar["waveform_encode_descr"] = (("beam_group",), np.array([descr]))
    return EchoData("EK80", {"Sonar": sonar, BEAM1: beam, "Vendor_specific": vend, "Environment": env},
                    source_file=source_file)


def from_azfp_arrays(d, source_file="synthetic.azfp"):
    """AZFP EchoData (convert/set_groups_azfp.py:417-466,583-608,736-770)."""
    ch = list(d["channel"])
    pt = _time1(d["ping_time"])
    C, P, S = d["backscatter_r"].shape
    beam = Dataset(coords={"channel": ch, "ping_time": pt, "range_sample": np.arange(S)})
    beam["backscatter_r"] = (("channel", "ping_time", "range_sample"), d["backscatter_r"])
    beam["transmit_duration_nominal"] = (("channel", "ping_time"), np.asarray(d["transmit_duration_nominal"], float))
    beam["frequency_nominal"] = (("channel",), np.asarray(d["frequency_nominal"], float))
    beam["equivalent_beam_angle"] = (("channel",), np.asarray(d["equivalent_beam_angle"], float))
    vend = Dataset(coords={"channel": ch})
    for k in ("number_of_samples_per_average_bin", "digitization_rate", "lock_out_index", "EL", "DS", "TVR",
              "VTX0", "Sv_offset"):
        vend[k] = (("channel",), np.asarray(d[k], float))
    env = Dataset(coords={"time1": pt})
    env["temperature"] = (("time1",), np.asarray(d["temperature"], float))
    return EchoData("AZFP", {BEAM1: beam, "Vendor_specific": vend, "Environment": env}, source_file=source_file)
