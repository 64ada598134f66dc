"""Independent files on HIP streams of their own (bench.py --tile-streams): the reference's two calls --
and the three of the chain -- issued under alternating ``torch.cuda.stream`` contexts, their kernels side by side on the
GPU, give what the same calls give one after the other on the default stream.  Every call launches on torch's current
stream and reads its results under it; what the library shares between calls (upload staging, the download stream,
stream-ordered scratch) must not leak from one stream to the other."""
import logging

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd as ep
    from echopype_amd import synth

    return torch, ep, synth


def _file(ep, synth, C, P, S, seed, ping0):
    d = synth.ek60_numpy(C, 4, 8)
    h = synth.ek60_params(C, P, ping0=ping0, ss_every=1)
    for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
              "absorption_indicative"):
        d[k] = h[k]
    d["ping_time"] = h["ping_time"]
    d["backscatter_r"] = ep.DeviceArray(synth.ek60_device(C, P, S, seed=seed, ss_every=1, ping0=ping0)["backscatter_r"])
    return ep.echodata.from_ek60_arrays(d, source_file=f"synthetic_{seed}.raw").to_device()


def _two_calls(ep, ed, dtype):
    ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
    return ds, ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_files_on_alternating_streams_equal_files_one_after_the_other(env, dtype):
    torch, ep, synth = env
    C, P, S = 4, 40000, 2000          # ~2 ms of kernel per file: launches issued 1 ms apart overlap
    files = [_file(ep, synth, C, P, S, seed=100 + i, ping0=i * P) for i in range(4)]
    logging.disable(logging.WARNING)
    try:
        ref = []
        for ed in files:
            ds, mv = _two_calls(ep, ed, dtype)
            ref.append((ds["Sv"].values.copy(), mv["Sv"].values.copy(), ds["echo_range"].values[:, :3].copy()))
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(2)]
        for rep in range(3):
            pending = []
            for i, ed in enumerate(files):
                st = streams[(i + rep) % 2]
                with torch.cuda.stream(st):
                    pending.append((_two_calls(ep, ed, dtype), st))      # nothing is read here: no call waits for the GPU
            for ((ds, mv), st), (sv0, mv0, rg0) in zip(pending, ref):
                with torch.cuda.stream(st):
                    sv, m, rg = ds["Sv"].values, mv["Sv"].values, ds["echo_range"].values[:, :3]
                np.testing.assert_array_equal(sv, sv0)                   # per-sample values: bit for bit
                np.testing.assert_array_equal(rg, rg0)
                np.testing.assert_array_equal(np.isnan(m), np.isnan(mv0))
                # (LDS atomics add in another order: a mean of 1.0000000000005 in linear units is 2e-12 dB, hence the absolute term)
                np.testing.assert_allclose(m, mv0, rtol=1e-11 if dtype == "float64" else 1e-5,
                                           atol=1e-11 if dtype == "float64" else 1e-4, equal_nan=True)
    finally:
        logging.disable(logging.NOTSET)


def test_chain_on_alternating_streams(env):
    torch, ep, synth = env
    C, P, S = 2, 30000, 2000
    files = [_file(ep, synth, C, P, S, seed=300 + i, ping0=i * P) for i in range(3)]
    logging.disable(logging.WARNING)
    try:
        def run(ed):
            ds = ep.calibrate.compute_Sv(ed, dtype="float64")
            ds = ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50)
            mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
            return ds, mv
        ref = []
        for ed in files:
            ds, mv = run(ed)
            ref.append((ds["Sv_corrected"].values.copy(), ds["Sv_noise"].values.copy(), mv["Sv"].values.copy()))
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(2)]
        pending = []
        for i, ed in enumerate(files):
            with torch.cuda.stream(streams[i % 2]):
                pending.append((run(ed), streams[i % 2]))
        for ((ds, mv), st), (c0, n0, m0) in zip(pending, ref):
            with torch.cuda.stream(st):
                c, n, m = ds["Sv_corrected"].values, ds["Sv_noise"].values, mv["Sv"].values
            np.testing.assert_array_equal(np.isnan(c), np.isnan(c0))
            np.testing.assert_allclose(c, c0, rtol=1e-11, atol=1e-11, equal_nan=True)
            np.testing.assert_allclose(n, n0, rtol=1e-11, atol=1e-11, equal_nan=True)
            np.testing.assert_allclose(m, m0, rtol=1e-11, atol=1e-11, equal_nan=True)
    finally:
        logging.disable(logging.NOTSET)
