"""Synthetic known-answer fixtures RESTATED from the reference's own test-suite (data
generators only -- no reference code is executed or copied):

  /root/reference/echopype/tests/mock_data.py:17-24,28-85,88-214      (mock Sv datasets, brute-force MVBS)
  /root/reference/echopype/tests/commongrid/conftest.py:28-98,121-166  (NaN positions, Sv sample)
  /root/reference/echopype/tests/clean/test_noise.py:902-987          (noise toy + seed(1) case)
  /root/reference/echopype/tests/calibrate/test_cal_params.py:57-77,751-868 (pulse-length tables)

Used by the CPU oracle KATs and by the GPU parity tests, so both are held to the same
known answers.  The reference leaves the ping-time jitter unseeded (mock_data.py:22); a
fixed seed is used here.
"""
import numpy as np
import pandas as pd

MOCK_NAN_ILOCS = [
    (1, 1, 10), (1, 0, 16), (0, 3, 6), (0, 2, 11), (0, 2, 6), (1, 1, 14), (0, 1, 17),
    (1, 4, 19), (0, 3, 3), (0, 0, 19), (0, 1, 5), (1, 2, 9), (1, 4, 18), (0, 1, 5),
    (0, 4, 4), (0, 1, 6), (1, 2, 2), (0, 1, 2), (0, 4, 8), (0, 1, 1),
]


def gen_ping_time(n, interval, jitter_ms=0, seed=7):
    t = pd.Timestamp("2018-07-01") + pd.to_timedelta(np.arange(n) * pd.to_timedelta(interval))
    if jitter_ms:
        jit = np.random.default_rng(seed).integers(jitter_ms, size=n)
        t = (t + pd.to_timedelta(jit, unit="ms")).sort_values()
    return t.values.astype("datetime64[ns]")


def sv_regular(channel_len=2, depth_len=100, depth_interval=0.5, ping_time_len=600,
               ping_time_interval="0.3s", jitter_ms=0, rng=None):
    rng = rng or np.random.default_rng(11)
    er = np.tile(np.arange(depth_len) * depth_interval, (channel_len, ping_time_len, 1)).astype(float)
    return dict(Sv=rng.random((channel_len, ping_time_len, depth_len)), echo_range=er,
                ping_time=gen_ping_time(ping_time_len, ping_time_interval, jitter_ms))


def sv_irregular(channel_len=2, depth_len=100, depth_interval=(0.5, 0.32, 0.13),
                 depth_ping_time_len=(100, 300, 200), ping_time_len=600,
                 ping_time_interval="0.3s", jitter_ms=0, rng=None):
    rng = rng or np.random.default_rng(12)
    assert sum(depth_ping_time_len) == ping_time_len
    parts = [np.tile(np.arange(depth_len) * d, (channel_len, n, 1)) for d, n in
             zip(depth_interval, depth_ping_time_len)]
    return dict(Sv=rng.random((channel_len, ping_time_len, depth_len)),
                echo_range=np.concatenate(parts, axis=1).astype(float),
                ping_time=gen_ping_time(ping_time_len, ping_time_interval, jitter_ms))


def mock_small(kind):
    """commongrid/conftest.py mock_Sv_dataset_{regular,irregular}: (2,10,20), Sv=linspace(0,1,20)."""
    sample = np.tile(np.linspace(0, 1, 20), (2, 10, 1))
    if kind == "regular":
        d = sv_regular(2, 20, 0.5, 10, "0.3s")
        d["Sv"] = sample.copy()
    else:
        d = sv_irregular(2, 20, (0.5, 0.32, 0.2), (2, 3, 5), 10, "0.3s", jitter_ms=30)
        d["Sv"] = sample.copy()
    # add_depth(depth_offset=2.5), tilt 0 -- applied BEFORE the NaNs are sprinkled
    # (conftest.py:152-165), so depth keeps valid coordinates where Sv is NaN
    d["depth"] = d["echo_range"] + 2.5
    if kind != "regular":
        for pos in MOCK_NAN_ILOCS:
            d["echo_range"][pos] = np.nan
            d["Sv"][pos] = np.nan
    return d


def brute_force_mvbs(d, ping_time_bin, range_bin, range_key="echo_range"):
    """mock_data.py:28-85 restated: triple loop, label slices inclusive at both ends,
    range edges arange(0, max+2, bin), plain np.mean of the selected linear values."""
    pt = pd.DatetimeIndex(d["ping_time"])
    idx = pd.Series(0, index=pt).resample(ping_time_bin).first().index
    p_edges = idx.union([idx[-1] + pd.Timedelta(ping_time_bin)]).values
    r_edges = np.arange(0, np.nanmax(d[range_key]) + 2, range_bin)
    lin = 10 ** (d["Sv"] / 10)
    C = d["Sv"].shape[0]
    out = np.full((C, len(p_edges) - 1, len(r_edges) - 1), np.nan)
    ptv = d["ping_time"]
    for c in range(C):
        for i in range(len(p_edges) - 1):
            psel = (ptv >= p_edges[i]) & (ptv <= p_edges[i + 1])
            er = d[range_key][c][psel]
            for j in range(len(r_edges) - 1):
                act = (er >= r_edges[j]) & (er < r_edges[j + 1])
                vals = lin[c][psel][act]
                out[c, i, j] = np.nan if vals.size == 0 else np.mean(vals)
    with np.errstate(divide="ignore", invalid="ignore"):
        return 10 * np.log10(out)


def noise_toy():
    """test_noise.py:905-940: ones with -30 at samples 30 and 60; echo_range linspace(0,10)."""
    data = np.ones(100)
    data[30] = -30
    data[60] = -30
    Sv = np.array([[data] * 10])
    er = np.array([[np.linspace(0, 10, 100)] * 10])
    return Sv, er, 0.001


def noise_seed1():
    """test_noise.py:953-980: np.random.seed(1) normal(-100, 2), echo_range linspace(0,3)."""
    st = np.random.RandomState(1)  # == np.random.seed(1); np.random.normal(...)
    Sv = st.normal(loc=-100, scale=2, size=(1, 10, 100))
    er = np.array([[np.linspace(0, 3, 100)] * 10])
    return Sv, er, 0.001


PULSE_TABLE = dict(
    pulse_length=np.array([[64, 128, 256, 512], [128, 256, 512, 1024]], float),
    table=np.array([[10, 20, 30, 40], [110, 120, 130, 140]], float),
)
PULSE_CASES = [  # (tau (C,P), expected (C,P)) -- test_cal_params.py:751-868
    (np.array([[64, 256, 128, 512], [512, 1024, 256, 128]], float),
     np.array([[10, 30, 20, 40], [130, 140, 120, 110]], float)),
    (np.array([[64, np.nan, 128, 512], [512, 1024, 256, np.nan]], float),
     np.array([[10, np.nan, 20, 40], [130, 140, 120, np.nan]], float)),
]
