"""Seeded synthetic echograms of the BASELINE.json shapes (SURVEY.md 8d recipe).

Two flavours of the same recipe:
  * ``ek60_numpy`` / ``ek80_numpy`` / ``azfp_numpy`` -- host arrays for the parity tests (the
    oracle and the HIP path consume the same arrays);
  * ``ek60_device`` -- generated directly in HBM with torch for the full-size bench volumes
    (16-130 GB never cross PCIe).
Raw EK60 power: float32(int16 ~ U[-12000, -2000]) * float32(10*log10(2)/256)
(convert/parse_base.py:24,302); 10 % of pings carry a NaN tail over the last 5 % of the range.
"""
import numpy as np

INDEX2POWER = np.float32(10.0 * np.log10(2.0) / 256.0)

EK60_FREQ = np.array([18e3, 38e3, 120e3, 200e3])
EK60_PT = np.array([2000.0, 2000.0, 250.0, 150.0])
EK60_G = np.array([22.9, 26.5, 27.0, 27.0])
EK60_PSI = np.array([-17.0, -20.6, -20.4, -20.2])
PULSE_LENGTHS = np.array([256e-6, 512e-6, 1024e-6, 2048e-6, 4096e-6])
T0 = np.datetime64("2026-05-01T00:00:00", "ns")


def fg_absorption(f_hz, T=10.0, S=35.0, P=10.0, pH=8.0):
    """Francois & Garrison absorption [dB/m] for the synthetic Environment group (host prep)."""
    from .utils.uwa import calc_absorption

    return calc_absorption(frequency=f_hz, temperature=T, salinity=S, pressure=P, pH=pH,
                           formula_source="FG")


def _channels(C):
    if C == 2:
        return np.array([1, 2])  # 38 / 120 kHz (cfg1)
    if C <= 4:
        return np.arange(C)
    return np.arange(C) % 4


def ek60_params(C, P, vary_tau=False, seed=0, ping0=0, ss_every=1):
    """Per-channel / per-ping parameters of the synthetic EK60 file (host, O(C*P)).  ``ping0``: global index of
    the first ping when the arrays are a ping shard / tile of a longer file (ping times and the sound-speed drift
    follow the global index).  ``ss_every``: the sound speed an EK60 records with every ping is the operator's
    setting -- it follows the slow drift in steps of this many pings (1: a new value every ping, the hardest case
    for the kernels that cache per-range-column terms)."""
    ch = _channels(C)
    p = np.arange(P) + ping0
    si = np.full((C, P), 2.56e-4)
    tau = np.full((C, P), 1.024e-3)
    if vary_tau:  # exercise the pulse-length table lookup (+ a NaN ping)
        rng = np.random.default_rng(seed + 99)
        tau = PULSE_LENGTHS[rng.integers(0, 5, size=(C, P))]
        tau[:, 0] = 1.024e-3
    ss = np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * (p // ss_every * ss_every) / 1e5), (C, 1))
    g = EK60_G[ch]
    return dict(
        channel=[f"GPT {int(EK60_FREQ[i] / 1e3)} kHz 00907205{i:04d} 1 ES{int(EK60_FREQ[i] / 1e3)}" for i in ch],
        frequency_nominal=EK60_FREQ[ch].copy(),
        sample_interval=si,
        transmit_duration_nominal=tau,
        transmit_power=np.tile(EK60_PT[ch][:, None], (1, P)),
        sound_speed_indicative=ss,
        absorption_indicative=np.tile(fg_absorption(EK60_FREQ[ch])[:, None], (1, P)),
        equivalent_beam_angle=EK60_PSI[ch].copy(),
        pulse_length=np.tile(PULSE_LENGTHS, (C, 1)),
        gain_correction=np.stack([g - 1.0, g - 0.5, g, g + 0.2, g + 0.3], axis=1),
        sa_correction=np.tile(np.array([-0.7, -0.6, -0.5, -0.3, -0.3]), (C, 1)),
        ping_time=T0 + (p * 1_000_000_000).astype("timedelta64[ns]"),
    )


def ek60_numpy(C=2, P=200, S=1000, seed=20260501, vary_tau=False, ss_every=1):
    """Host arrays: backscatter_r f32 (C,P,S) + params."""
    rng = np.random.default_rng(seed)
    raw = rng.integers(-12000, -2000, size=(C, P, S), dtype=np.int16).astype(np.float32) * INDEX2POWER
    nan_pings = rng.random(P) < 0.10
    tail = max(1, int(round(0.05 * S)))
    raw[:, nan_pings, S - tail:] = np.nan
    d = ek60_params(C, P, vary_tau=vary_tau, seed=seed, ss_every=ss_every)
    d["backscatter_r"] = raw
    return d


def ek60_device(C, P, S, seed=20260501, device=None, chunk_pings=20000, ping0=0, ss_every=2000):
    """Same recipe generated in HBM: returns dict of torch CUDA tensors (raw f32 + f64 params).  The recorded sound
    speed changes every ``ss_every`` pings (see :func:`ek60_params`)."""
    import torch

    dev = device or torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    raw = torch.empty((C, P, S), dtype=torch.float32, device=dev)
    tail = max(1, int(round(0.05 * S)))
    for c in range(C):
        for p0 in range(0, P, chunk_pings):
            p1 = min(P, p0 + chunk_pings)
            blk = torch.randint(-12000, -2000, (p1 - p0, S), generator=g, device=dev, dtype=torch.int16)
            raw[c, p0:p1] = blk.to(torch.float32) * float(INDEX2POWER)
            del blk
    nan_pings = torch.rand(P, generator=g, device=dev) < 0.10
    idx = torch.nonzero(nan_pings).flatten()
    raw[:, idx, S - tail:] = float("nan")
    h = ek60_params(C, P, ping0=ping0, ss_every=ss_every)
    out = {"backscatter_r": raw}
    for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
              "absorption_indicative", "equivalent_beam_angle", "frequency_nominal", "pulse_length",
              "gain_correction", "sa_correction"):
        out[k] = torch.from_numpy(np.ascontiguousarray(h[k], dtype=np.float64)).to(dev)
    out["ping_time_ns"] = torch.from_numpy(h["ping_time"].astype(np.int64)).to(dev)
    out["ping_time"] = h["ping_time"]
    out["channel"] = h["channel"]
    return out


def ek60_device_i16(C, P, S, seed=20260501, device=None, chunk_pings=20000, ss_every=2000):
    """The same recipe as :func:`ek60_device` before the converter's float conversion: int16 power
    samples (C,P,S) + the recorded length of every ping (C,P) int32."""
    import torch

    dev = device or torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    raw = torch.empty((C, P, S), dtype=torch.int16, device=dev)
    for c in range(C):
        for p0 in range(0, P, chunk_pings):
            p1 = min(P, p0 + chunk_pings)
            raw[c, p0:p1] = torch.randint(-12000, -2000, (p1 - p0, S), generator=g, device=dev, dtype=torch.int16)
    short = torch.rand(P, generator=g, device=dev) < 0.10
    n_valid = torch.full((C, P), S, dtype=torch.int32, device=dev)
    n_valid[:, short] = S - max(1, int(round(0.05 * S)))
    h = ek60_params(C, P, ss_every=ss_every)
    out = {"raw_i16": raw, "n_valid": n_valid}
    for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
              "absorption_indicative", "equivalent_beam_angle", "frequency_nominal", "pulse_length",
              "gain_correction", "sa_correction"):
        out[k] = torch.from_numpy(np.ascontiguousarray(h[k], dtype=np.float64)).to(dev)
    out["ping_time_ns"] = torch.from_numpy(h["ping_time"].astype(np.int64)).to(dev)
    out["ping_time"] = h["ping_time"]
    return out


# ---------------------------------------------------------------------------------------- EK80
def ek80_filters(seed=20260501):
    """Deterministic stand-ins for the Vendor_specific WBT/PC filter coefficients (complex64)."""
    k47, k91 = np.arange(47), np.arange(91)
    wbt = (np.hanning(47) * np.exp(2j * np.pi * 0.045 * k47) / 10).astype(np.complex64)
    pc = (np.hanning(91) * np.exp(2j * np.pi * 0.13 * k91) / 20).astype(np.complex64)
    return dict(wbt_fil=wbt, wbt_decifac=6, pc_fil=pc, pc_decifac=2)


EK80_BB = dict(
    frequency_nominal=np.array([70e3, 120e3]),
    f_start=np.array([45e3, 90e3]),
    f_stop=np.array([90e3, 170e3]),
    tau=np.array([1.024e-3, 0.512e-3]),
    transmit_power=np.array([750.0, 250.0]),
    z_er=np.array([5400.0, 5400.0]),
    z_et=np.array([75.0, 75.0]),
    psi=np.array([-20.7, -20.7]),
    gain=np.array([27.0, 26.8]),
    sa=np.array([-0.1, -0.05]),
    angle_offset_alongship=np.array([0.05, -0.03]),
    angle_offset_athwartship=np.array([-0.02, 0.04]),
    beamwidth_alongship=np.array([6.8, 6.6]),
    beamwidth_athwartship=np.array([6.9, 6.5]),
)


def ek80_numpy(C=2, P=16, S=1024, B=4, seed=20260504, waveform="BB", replicas=None,
               mixed_nan=False):
    """Host arrays for EK80 complex data: backscatter_r/_i f64 (C,P,S,B) + params.

    Complex noise N(0,1)+iN(0,1) * 1e-3 plus replica-shaped echoes at 3 random ranges per ping,
    NaN tail on ~10 % of pings, optional per-sector (mixed) NaNs to exercise the per-sector path.
    """
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((C, P, S, B)) + 1j * rng.standard_normal((C, P, S, B))) * 1e-3
    if replicas is not None:
        for c in range(C):
            r = replicas[c]
            for p in range(P):
                for start in rng.integers(0, max(1, S - 8), size=3):
                    n = min(r.size, S - start)
                    amp = 0.2 + 0.6 * rng.random()
                    x[c, p, start:start + n, :] += amp * r[:n, None] * np.exp(1j * rng.random(B))[None, :]
    re = np.ascontiguousarray(x.real, dtype=np.float64)
    im = np.ascontiguousarray(x.imag, dtype=np.float64)
    nan_pings = rng.random(P) < 0.10
    nan_pings[min(1, P - 1)] = True
    tail = max(1, int(round(0.05 * S)))
    re[:, nan_pings, S - tail:, :] = np.nan
    im[:, nan_pings, S - tail:, :] = np.nan
    if mixed_nan:
        re[0, 0, S // 3: S // 3 + 5, 1] = np.nan
        im[0, 0, S // 3: S // 3 + 5, 1] = np.nan
        im[C - 1, P - 1, 10, B - 1] = np.nan   # imag-only NaN (convert sets imag 0 -> NaN)
        re[C - 1, P - 1, 17, :] = np.nan       # whole sample NaN in the middle of a ping
        im[C - 1, P - 1, 17, :] = np.nan
        # beam 0 missing, the other sectors valid: echo_range is masked there (range.py:143-148), hence Sv NaN
        re[0, min(1, P - 1), S // 2: S // 2 + 3, 0] = np.nan
    p = np.arange(P)
    d = dict(EK80_BB)
    d = {k: (v[:C].copy() if isinstance(v, np.ndarray) else v) for k, v in d.items()}
    d.update(
        backscatter_r=re, backscatter_i=im,
        channel=[f"WBT 4000{i}-15 ES{int(d['frequency_nominal'][i] / 1e3)}-7C" for i in range(C)],
        sample_interval=np.full((C, P), 8e-6),
        sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1)),
        slope=np.full(C, 0.05), fs=np.full(C, 1.5e6),
        ping_time=T0 + (p * 1_000_000_000).astype("timedelta64[ns]"),
        waveform=waveform,
    )
    return d


# ---------------------------------------------------------------------------------------- AZFP
def azfp_numpy(C=4, P=60, S=500, seed=20260507):
    rng = np.random.default_rng(seed)
    counts = rng.integers(2000, 60000, size=(C, P, S)).astype(np.float32)
    f = np.array([38e3, 125e3, 200e3, 455e3])[:C]
    p = np.arange(P)
    return dict(
        backscatter_r=counts, frequency_nominal=f,
        channel=[f"55030-{int(x / 1e3)}-1" for x in f],
        transmit_duration_nominal=np.tile(np.array([5e-4, 3e-4, 3e-4, 1.5e-4])[:C, None], (1, P)),
        number_of_samples_per_average_bin=np.array([20.0, 10.0, 10.0, 5.0])[:C],
        digitization_rate=np.full(C, 64000.0), lock_out_index=np.array([0.0, 2.0, 2.0, 4.0])[:C],
        EL=np.array([142.8, 144.0, 141.5, 140.3])[:C], DS=np.array([0.02293, 0.02243, 0.02273, 0.02293])[:C],
        TVR=np.array([169.9, 170.3, 172.6, 175.5])[:C], VTX0=np.array([105.2, 117.8, 110.1, 63.8])[:C],
        Sv_offset=np.array([1.1, 1.4, 1.4, 1.3])[:C],
        equivalent_beam_angle=10 ** (np.array([-11.8, -18.3, -18.5, -18.6])[:C] / 10),
        temperature=np.full(P, 8.5), salinity=29.6, pressure=60.0,
        ping_time=T0 + (p * 3_000_000_000).astype("timedelta64[ns]"),
    )
