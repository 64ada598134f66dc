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


# ---- the package's own loop: echopype_amd.pipeline -------------------------------------------------------------------------
@pytest.mark.parametrize("streams,lag", [(2, 1), (3, 2), (0, 1), (2, 0)])
def test_pipeline_sv_mvbs_equals_the_plain_loop(env, streams, lag):
    """ep.pipeline.sv_mvbs: consecutive files on alternating side streams, results handed out ``lag`` files late, in
    order, assembled, and readable on the CALLER's stream (its stream waits for the file's on the device) -- the plain
    loop's datasets."""
    torch, ep, synth = env
    from echopype_amd.xr_lite import DeferredDataset

    C, P, S = 4, 30000, 2000
    files = [_file(ep, synth, C, P, S, seed=500 + i, ping0=i * P) for i in range(5)]
    logging.disable(logging.WARNING)
    try:
        ref = []
        for ed in files:
            ds, mv = _two_calls(ep, ed, "float64")
            ref.append((ds["Sv"].values.copy(), mv["Sv"].values.copy(), mv["ping_time"].values.copy()))
        torch.cuda.synchronize()
        launched = []

        def feed():
            for i, ed in enumerate(files):
                launched.append(i)
                yield ed

        got = 0
        for k, (ds, mv) in enumerate(ep.pipeline.sv_mvbs(feed(), streams=streams, lag=lag, range_bin="1m", ping_time_bin="20s")):
            assert len(launched) == min(len(files), k + lag + 1)        # handed out ``lag`` files late
            assert not isinstance(mv, DeferredDataset) or mv.resolved   # ... assembled
            sv0, mv0, t0 = ref[k]
            # read on the caller's stream, no stream context: its stream was made to wait for the file's
            np.testing.assert_array_equal(ds["Sv"].data.tensor.cpu().numpy(), sv0)
            np.testing.assert_array_equal(mv["ping_time"].values, t0)
            np.testing.assert_allclose(mv["Sv"].values, mv0, rtol=1e-11, atol=1e-11, equal_nan=True)
            got += 1
        assert got == len(files)
    finally:
        logging.disable(logging.NOTSET)


def test_pipeline_run_orders_items_behind_the_callers_stream(env):
    """What the caller queued on its stream before an item is launched is finished before the item's kernels read it."""
    torch, ep, synth = env

    base = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")

    def items():
        for i in range(6):
            base.add_(1.0)              # on the caller's stream, right before the item is launched
            yield i

    def fn(i):
        return base.sum()               # on the item's side stream

    # (lag 0: an item is handed out -- the caller's stream waits for it -- before the caller touches ``base`` again)
    out = [float(t.item()) for t in ep.pipeline.run(items(), fn, streams=2, lag=0, settle=False)]
    assert out == [float((i + 1) * (1 << 22)) for i in range(6)]


def test_pipeline_streams_were_seen_running_side_by_side(env):
    """The side streams the pipeline deals items to are checked, not assumed: two streams the HIP runtime has bound to
    one hardware queue run their kernels in turns (the headline's two-stream gain came and went with that between runs
    of one bench process, profiles/r06_stream_pairs.txt).  A stream against itself is the serial case."""
    torch, ep, _ = env
    a, b = ep.pipeline._Streams.get(2)
    assert a is not b and ep.pipeline._Streams.checked[(a.device, 2)]
    assert ep.pipeline._runs_beside(a, b)
    assert not ep.pipeline._runs_beside(a, a)
    # more streams than the runtime has hardware queues for (GPU_MAX_HW_QUEUES, 8 unless the user set it): the check sets
    # a sharing candidate aside, so a set of three still runs pairwise side by side
    three = ep.pipeline._Streams.get(3)
    assert len({id(s) for s in three}) == 3
    if ep.pipeline._Streams.checked[(a.device, 3)]:
        assert all(ep.pipeline._runs_beside(three[i], three[j]) for i in range(3) for j in range(i + 1, 3))
    torch.cuda.synchronize()
