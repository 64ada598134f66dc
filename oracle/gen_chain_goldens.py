#!/usr/bin/env python3
"""Generate tests/golden/ref_chain_goldens.npz by EXECUTING THE REFERENCE'S OWN array-level methods
of the calibration chain (authoring container only, needs /root/reference):

  calibrate/range.py::compute_range_EK, range_mod_TVG_EK, compute_range_AZFP
  calibrate/calibrate_ek.py::CalibrateEK._cal_power_samples       (EK60 Sv / TS; EK80 power with a GPT channel)
  calibrate/calibrate_azfp.py::CalibrateAZFP._cal_power_samples   (AZFP Sv / TS)
  calibrate/calibrate_ek.py::CalibrateEK80._cal_complex_samples   (EK80 complex, BB and CW, Sv / TS) with
      _get_power_from_complex, _get_B_theta_phi_m and ek80_complex.py::compress_pulse, get_norm_fac,
      get_tau_effective; the transmit replicas come from tapered_chirp + filter_decimate_chirp (the two
      steps get_transmit_signal chains after reading the file's parameters)

xarray cannot be installed here; those method bodies are arithmetic on labelled arrays, so they are
run over oracle/xr_shim.py (broadcast by dimension name, element-wise ufuncs, where / isnull / isel /
transpose -- strict: anything unimplemented raises).  The calibrator objects are created without
their constructors (which parse files and parameters through the EchoData / ECS machinery) and given
exactly the attributes the methods read; the parameter-selection code has its own goldens / KATs.
Output = data only: seeded inputs + the outputs of the reference's code.

A second file, tests/golden/ref_seam_goldens.npz, holds two LONG cases whose range axis crosses the seams of the HIP
kernels (round-3 review, item 3): EK60 power with S = 2052 (two 1024-sample chunk boundaries of the power kernels and a
4-sample tail) and EK80 broadband with S = 4200 (the overlap-save tiles of csrc/ek80_fft.hip: echoes straddling samples
1872, 2048, 3744 and 4096 +- taps, partly-NaN sectors inside the overlap regions, a NaN in beam 0 there); inputs are
float32-representable and stored as float32 to keep the file under 2 MB.
"""
import logging
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xr_shim  # noqa: E402
from gen_goldens import REF, _load  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_chain_goldens.npz")
OUT_SEAM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_seam_goldens.npz")
DA, DS = xr_shim.DataArray, xr_shim.Dataset


def load_reference_calibrators():
    xr = types.ModuleType("xarray")
    xr.DataArray, xr.Dataset, xr.where, xr.merge = DA, DS, xr_shim.where, xr_shim.merge
    xr.apply_ufunc = xr_shim.apply_ufunc
    sys.modules["xarray"] = xr

    def pkg(name, path=()):
        m = types.ModuleType(name)
        m.__path__ = list(path)
        sys.modules[name] = m
        return m

    pkg("echopype", [REF]); pkg("echopype.calibrate", [f"{REF}/calibrate"]); pkg("echopype.convert")
    pkg("echopype.utils")
    ed = pkg("echopype.echodata")
    ed.EchoData = object
    sim = types.ModuleType("echopype.echodata.simrad"); sim.retrieve_correct_beam_group = None
    sys.modules[sim.__name__] = sim
    log = types.ModuleType("echopype.utils.log"); log._init_logger = logging.getLogger
    sys.modules[log.__name__] = log
    sg = types.ModuleType("echopype.convert.set_groups_ek80")
    sg.DECIMATION, sg.FILTER_IMAG, sg.FILTER_REAL = "deci_fac", "coeffs_imag", "coeffs_real"
    sys.modules[sg.__name__] = sg
    for name, attrs in (("ecs", ["ECSParser", "conform_channel_order", "ecs_ds2dict", "ecs_ev2ep"]),
                        ("cal_params", ["_get_interp_da", "get_cal_params_EK", "get_cal_params_AZFP"]),
                        ("env_params", ["get_env_params_EK", "get_env_params_AZFP"])):
        m = types.ModuleType(f"echopype.calibrate.{name}")
        for a in attrs:
            setattr(m, a, None)
        sys.modules[m.__name__] = m
    # range.py::compute_range_AZFP harmonises sound_speed to ping_time; the generator supplies it per ping already
    sys.modules["echopype.calibrate.env_params"].harmonize_env_param_time = lambda p, ping_time: p
    ekc = _load("echopype.calibrate.ek80_complex", f"{REF}/calibrate/ek80_complex.py")
    _load("echopype.calibrate.calibrate_base", f"{REF}/calibrate/calibrate_base.py")
    rng_mod = _load("echopype.calibrate.range", f"{REF}/calibrate/range.py")
    ek = _load("echopype.calibrate.calibrate_ek", f"{REF}/calibrate/calibrate_ek.py")
    az = _load("echopype.calibrate.calibrate_azfp", f"{REF}/calibrate/calibrate_azfp.py")
    return rng_mod, ek, az, ekc


class _ED:
    def __init__(self, sonar_model, groups):
        self.sonar_model, self._g = sonar_model, groups

    def __getitem__(self, k):
        return self._g[k]


def _cp(a, chans, pings):
    return DA(a, {"channel": chans, "ping_time": pings}, ["channel", "ping_time"])


def ek_case(ek, rng_mod, g, tag, sonar, C, P, S, seed, gpt=None, psi_per_ping=False):
    rng = np.random.default_rng(seed)
    chans = np.array([f"ch{i}" for i in range(C)])
    pings = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    raw = (rng.integers(-12000, -2000, (C, P, S)).astype(np.float32) * np.float32(10 * np.log10(2) / 256))
    raw[:, 1, S - 7:] = np.nan
    raw[0, 3, :] = np.nan
    si = np.full((C, P), 2.56e-4) * (1 + 0.5 * (np.arange(C) % 2))[:, None]
    tau = np.full((C, P), 1.024e-3)
    tau[-1, :] = 0.512e-3
    pt = np.array([2000.0, 1000.0, 250.0, 150.0])[:C, None] * np.ones((1, P))
    f = np.array([18e3, 38e3, 120e3, 200e3])[:C]
    c_w = 1490.0 + 3.0 * rng.random((C, P))
    alpha = np.array([0.0021, 0.0098, 0.0374, 0.0527])[:C, None] * (1 + 0.01 * rng.random((C, P)))
    gain = np.array([22.9, 26.5, 27.0, 25.3])[:C, None] + 0.2 * rng.random((C, P))
    sa = -0.6 + 0.2 * rng.random((C, P))
    psi = np.array([-17.0, -20.6, -20.4, -20.2])[:C]
    if psi_per_ping:  # a (channel, ping_time) cal parameter: the reference broadcasts it into CSv (calibrate_ek.py:154-162)
        psi = psi[:, None] + 0.3 * rng.random((C, P))
    beam = DS(coords={"channel": chans, "ping_time": pings, "range_sample": np.arange(S)})
    beam["backscatter_r"] = DA(raw, dims=["channel", "ping_time", "range_sample"])
    beam["sample_interval"] = _cp(si, chans, pings)
    beam["transmit_duration_nominal"] = _cp(tau, chans, pings)
    beam["transmit_power"] = _cp(pt, chans, pings)
    beam["frequency_nominal"] = DA(f, {"channel": chans}, ["channel"])
    vend = DS(coords={"channel": chans})
    if sonar == "EK80":
        vend["transceiver_type"] = DA(np.array(gpt), {"channel": chans}, ["channel"])
    env = {"sound_speed": _cp(c_w, chans, pings), "sound_absorption": _cp(alpha, chans, pings)}
    cal = {"gain_correction": _cp(gain, chans, pings), "sa_correction": _cp(sa, chans, pings),
           "equivalent_beam_angle": _cp(psi, chans, pings) if psi_per_ping else DA(psi, {"channel": chans}, ["channel"])}
    cls = ek.CalibrateEK60 if sonar == "EK60" else ek.CalibrateEK80
    obj = object.__new__(cls)
    obj.echodata = _ED(sonar, {})
    obj.sonar_type = sonar
    obj.beam, obj.vend, obj.env_params, obj.cal_params = beam, vend, env, cal
    obj.waveform_mode, obj.encode_mode = "CW", "power"
    obj.range_meter = rng_mod.compute_range_EK(sonar, beam, env)
    for k, v in dict(raw=raw, sample_interval=si, tau=tau, transmit_power=pt, frequency=f, sound_speed=c_w,
                     absorption=alpha, gain=gain, sa=sa, psi=psi).items():
        g[f"{tag}_{k}"] = v
    if gpt is not None:
        g[f"{tag}_is_gpt"] = np.array([t == "GPT" for t in gpt])
    g[f"{tag}_echo_range"] = obj.range_meter.transpose("channel", "ping_time", "range_sample").data
    for cal_type in ("Sv", "TS"):
        out = obj._cal_power_samples(cal_type)  # the tx-signal attempt fails inside its own try/except -> nominal tau
        g[f"{tag}_{cal_type}"] = out[cal_type].transpose("channel", "ping_time", "range_sample").data
        if cal_type == "Sv":
            g[f"{tag}_tau_effective"] = np.asarray(out["tau_effective"].data, dtype=np.float64)


def azfp_case(az, g, C, P, S, seed):
    rng = np.random.default_rng(seed)
    chans = np.array([f"ch{i}" for i in range(C)])
    pings = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(2, "s")
    counts = rng.integers(5000, 50000, (C, P, S)).astype(np.float64)
    counts[1, 2, 5:9] = np.nan
    ch = lambda a: DA(np.asarray(a, float)[:C], {"channel": chans}, ["channel"])  # noqa: E731
    beam = DS(coords={"channel": chans, "ping_time": pings, "range_sample": np.arange(S)})
    beam["backscatter_r"] = DA(counts, dims=["channel", "ping_time", "range_sample"])
    tau = np.array([3e-4, 5e-4, 1e-3])[:C]
    beam["transmit_duration_nominal"] = ch(tau)
    beam["frequency_nominal"] = ch([38e3, 125e3, 200e3])
    vend = DS(coords={"channel": chans})
    N, fdig, L = [2, 4, 8][:C], [64000.0, 64000.0, 20000.0][:C], [0, 10, 3][:C]
    vend["number_of_samples_per_average_bin"], vend["digitization_rate"], vend["lock_out_index"] = ch(N), ch(fdig), ch(L)
    c_w = 1480.0 + 2.0 * rng.random(P)
    alpha = np.array([0.0098, 0.0395, 0.0522])[:C]
    env = {"sound_speed": DA(c_w, {"ping_time": pings}, ["ping_time"]), "sound_absorption": ch(alpha)}
    calp = dict(EL=[142.8, 142.5, 142.0], DS=[0.0229, 0.0230, 0.0231], TVR=[170.9, 168.8, 169.4],
                VTX0=[105.0, 106.2, 109.7], equivalent_beam_angle=[0.0174, 0.0070, 0.0070],
                Sv_offset=[1.0, 1.2, 1.5])
    cal = {k: ch(v) for k, v in calp.items()}
    obj = object.__new__(az.CalibrateAZFP)
    obj.echodata = _ED("AZFP", {"Sonar/Beam_group1": beam, "Vendor_specific": vend})
    obj.sonar_type, obj.env_params, obj.cal_params = "AZFP", env, cal
    g.update({"azfp_counts": counts, "azfp_tau": tau, "azfp_N": np.array(N, float), "azfp_f": np.array(fdig),
              "azfp_L": np.array(L, float), "azfp_sound_speed": c_w, "azfp_absorption": alpha})
    for k, v in calp.items():
        g[f"azfp_{k}"] = np.array(v[:C])
    for cal_type in ("Sv", "TS"):
        out = obj._cal_power_samples(cal_type)
        g[f"azfp_{cal_type}"] = out[cal_type].transpose("channel", "ping_time", "range_sample").data
        g[f"azfp_echo_range_{cal_type}"] = out["echo_range"].transpose("channel", "ping_time", "range_sample").data


def ek80_complex_case(ek, ekc, rng_mod, g, tag, waveform, C, P, S, B, seed, seam=False):
    rng = np.random.default_rng(seed)
    chans = np.array([f"ch{i}" for i in range(C)])
    pings = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    k47, k91 = np.arange(47), np.arange(91)
    coeff = dict(wbt_fil=(np.hanning(47) * np.exp(2j * np.pi * 0.045 * k47) / 10).astype(np.complex64), wbt_decifac=6,
                 pc_fil=(np.hanning(91) * np.exp(2j * np.pi * 0.13 * k91) / 20).astype(np.complex64), pc_decifac=2)
    fs = 1.5e6
    tau = np.array([1.024e-3, 0.512e-3])[:C]
    f0 = np.array([45e3, 90e3])[:C] if waveform == "BB" else np.array([38e3, 120e3])[:C]
    f1 = np.array([90e3, 170e3])[:C] if waveform == "BB" else f0
    tx, tx_time = {}, {}
    for i, ch in enumerate(chans):  # what get_transmit_signal does per channel (ek80_complex.py:245-280)
        y, _ = ekc.tapered_chirp(fs, np.array([tau[i]]), np.array([0.05]), np.array([f0[i]]), np.array([f1[i]]))
        tx[str(ch)], tx_time[str(ch)] = ekc.filter_decimate_chirp(coeff, y, fs)
    x = (rng.standard_normal((C, P, S, B)) + 1j * rng.standard_normal((C, P, S, B))) * 1e-3
    for c in range(C):
        r = tx[str(chans[c])]
        for p in range(P):
            k0 = int(rng.integers(10, S - 10))
            n = min(r.size, S - k0)
            x[c, p, k0:k0 + n, :] += 0.3 * r[:n, None]
    if seam:  # echoes across the tile seams of the LDS-FFT kernel (2048-sample tiles, 1872 new samples each) ...
        for c in range(C):
            r = tx[str(chans[c])]
            for p in range(P):
                for k0 in (1872 - r.size // 2, 2048 - r.size // 3, 2048 + 3, 3744 - 5, 4096 - r.size // 2, S - r.size // 2):
                    n = min(r.size, S - k0)
                    x[c, p, k0:k0 + n, :] += (0.2 + 0.1 * p) * r[:n, None]
        # ... partly-NaN sectors inside the overlap regions, a NaN in beam 0 there (masks echo_range, range.py:143-148)
        x[1, 0, 1860:1866, B - 1] = np.nan
        x[0, 1, 2040:2052, 2] = np.nan
        x[1, 3, 3740:3750, 0] = np.nan
        x[0, 0, 2047:2049, :] = np.nan  # every sector missing on both sides of sample 2048
        x = x.real.astype(np.float32).astype(np.float64) + 1j * x.imag.astype(np.float32).astype(np.float64)
    x[:, 1, S - 9:, :] = np.nan
    x[0, 2] = np.nan
    # partly-NaN samples: some sectors missing (the reference zeroes them for the convolution, restores the NaN and
    # averages the remaining sectors -- a NaN-skipping complex mean, calibrate_ek.py:483 over xarray's default skipna);
    # a NaN in beam 0 also masks echo_range (range.py:143-148)
    x[1, 0, 40:44, B - 1] = np.nan
    x[0, 3, 100, 0] = np.nan
    x[1, 3, 7:9, 1:3] = np.nan
    si = np.full((C, P), 1.0 / (fs / 12))
    pt = np.array([750.0, 250.0])[:C, None] * np.ones((1, P))
    c_w = 1490.0 + 2.0 * rng.random((C, P))
    alpha = np.array([0.0212, 0.0395])[:C, None] * np.ones((1, P))
    gain = np.array([26.1, 27.3])[:C, None] + 0.1 * rng.random((C, P))
    sa = np.array([-0.4, -0.3])[:C, None] * np.ones((1, P))
    psi = np.array([-20.7, -20.4])[:C]
    fc = (f0 + f1) / 2
    dims4 = ["channel", "ping_time", "range_sample", "beam"]
    beam = DS(coords={"channel": chans, "ping_time": pings, "range_sample": np.arange(S), "beam": np.arange(B)})
    beam["backscatter_r"] = DA(np.ascontiguousarray(x.real), dims=dims4)
    beam["backscatter_i"] = DA(np.ascontiguousarray(x.imag), dims=dims4)
    beam["sample_interval"] = _cp(si, chans, pings)
    beam["transmit_duration_nominal"] = _cp(np.tile(tau[:, None], (1, P)), chans, pings)
    beam["transmit_power"] = _cp(pt, chans, pings)
    beam["frequency_nominal"] = DA((f0 + f1) / 2 if waveform == "CW" else np.array([70e3, 120e3])[:C], {"channel": chans},
                                   ["channel"])
    vend = DS(coords={"channel": chans})
    vend["transceiver_type"] = DA(np.array(["WBT"] * C), {"channel": chans}, ["channel"])
    chd = lambda a: DA(np.asarray(a, float)[:C], {"channel": chans}, ["channel"])  # noqa: E731
    env = {"sound_speed": _cp(c_w, chans, pings), "sound_absorption": _cp(alpha, chans, pings)}
    cal = {"gain_correction": _cp(gain, chans, pings), "sa_correction": _cp(sa, chans, pings),
           "equivalent_beam_angle": chd(psi), "impedance_transceiver": chd([5400.0, 5400.0]),
           "impedance_transducer": chd([75.0, 75.0]), "receiver_sampling_frequency": chd([fs, fs]),
           "angle_offset_alongship": chd([0.05, -0.11]), "angle_offset_athwartship": chd([-0.08, np.nan]),
           "beamwidth_alongship": chd([6.9, 6.6]), "beamwidth_athwartship": chd([7.1, 6.5])}
    obj = object.__new__(ek.CalibrateEK80)
    obj.echodata = _ED("EK80", {})
    obj.sonar_type = "EK80"
    obj.beam, obj.vend, obj.env_params, obj.cal_params = beam, vend, env, cal
    obj.waveform_mode, obj.encode_mode, obj.drop_last_hanning_zero = waveform, "complex", False
    obj.freq_center = chd(fc)
    obj.range_meter = rng_mod.compute_range_EK("EK80", beam, env)
    ek.get_filter_coeff = lambda vend: None                      # file parsing; the replicas are built above
    ek.get_transmit_signal = lambda *a, **k: (tx, tx_time)
    for k, v in dict(re=x.real, im=x.imag, sample_interval=si, tau=tau, transmit_power=pt, sound_speed=c_w,
                     absorption=alpha, gain=gain, sa=sa, psi=psi, f_center=fc).items():
        g[f"{tag}_{k}"] = v.astype(np.float32) if seam and k in ("re", "im") else v  # (exact: the values are float32's)
    for name in ("angle_offset_alongship", "angle_offset_athwartship", "beamwidth_alongship", "beamwidth_athwartship"):
        g[f"{tag}_{name}"] = cal[name].data
    for i, ch in enumerate(chans):
        g[f"{tag}_replica{i}"] = tx[str(ch)]
    g[f"{tag}_echo_range"] = obj.range_meter.transpose("channel", "ping_time", "range_sample").data
    for cal_type in ("Sv", "TS"):
        out = obj._cal_complex_samples(cal_type)
        g[f"{tag}_{cal_type}"] = np.asarray(out[cal_type].transpose("channel", "ping_time", "range_sample").data,
                                            dtype=np.float64)
        if cal_type == "Sv":
            g[f"{tag}_tau_effective"] = np.asarray(out["tau_effective"].data, dtype=np.float64)


def main():
    logging.disable(logging.WARNING)
    rng_mod, ek, az, ekc = load_reference_calibrators()
    g = {}
    ek_case(ek, rng_mod, g, "ek60", "EK60", 3, 6, 40, 1)
    ek_case(ek, rng_mod, g, "ek80p", "EK80", 3, 5, 32, 2, gpt=["WBT", "GPT", "WBT"])
    ek_case(ek, rng_mod, g, "ek60psi", "EK60", 3, 7, 36, 6, psi_per_ping=True)
    azfp_case(az, g, 3, 4, 24, 3)
    ek80_complex_case(ek, ekc, rng_mod, g, "ek80bb", "BB", 2, 4, 420, 4, 4)
    ek80_complex_case(ek, ekc, rng_mod, g, "ek80cw", "CW", 2, 4, 300, 4, 5)
    np.savez_compressed(OUT, **g)
    print("wrote", os.path.normpath(OUT), f"{os.path.getsize(OUT)/1024:.1f} KiB,", len(g), "arrays")
    for k in ("ek60_Sv", "ek60_TS", "ek80p_Sv", "azfp_Sv", "azfp_TS", "ek80bb_Sv", "ek80bb_TS", "ek80cw_Sv"):
        print(k, g[k].shape, "NaN:", int(np.isnan(g[k]).sum()), "mean:", float(np.nanmean(g[k])))
    # the long cases across the kernels' seams
    h = {}
    ek_case(ek, rng_mod, h, "ek60seam", "EK60", 2, 4, 2052, 7)
    ek80_complex_case(ek, ekc, rng_mod, h, "ek80bbseam", "BB", 2, 4, 4200, 4, 8, seam=True)
    np.savez_compressed(OUT_SEAM, **h)
    print("wrote", os.path.normpath(OUT_SEAM), f"{os.path.getsize(OUT_SEAM)/1024:.1f} KiB,", len(h), "arrays")
    for k in ("ek60seam_Sv", "ek60seam_TS", "ek80bbseam_Sv", "ek80bbseam_TS"):
        print(k, h[k].shape, "NaN:", int(np.isnan(h[k]).sum()), "mean:", float(np.nanmean(h[k])))


if __name__ == "__main__":
    main()
