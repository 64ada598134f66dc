"""The reference's two calls without a wait for the GPU (round-4 review, item 1): ``compute_Sv(echodata)`` then
``compute_MVBS(ds_Sv)`` on device-resident echodata launch their kernels and return -- parameter uploads go through a
side stream (ops._Uploader), the size of the range grid is bounded on the host, the {nanmin, nanmax, NaN count} the grid
needs travel back on a download stream (ops.fetch_async) and the MVBS dataset is assembled on first use
(xr_lite.DeferredDataset).  Results are the ones the eager route gives."""
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

    return torch, ep


def test_uploader_and_fetch_async_round_trip_behind_a_busy_stream(env):
    """Arrays of several dtypes / sizes go up while a long kernel occupies the compute stream; a kernel launched after
    the upload sees the data (device-side event wait); the staging buffers are re-used; fetch_async returns what a
    blocking .cpu() would, and does not wait for work queued after it."""
    torch, ep = env
    from echopype_amd import ops

    rng = np.random.default_rng(0)
    busy = torch.empty(1 << 28, dtype=torch.float32, device="cuda")  # 1 GiB: a fill takes a fraction of a millisecond
    for rep in range(3):
        arrs = [rng.standard_normal(n) for n in (1, 5, 1000, 250_001)] + [rng.integers(0, 9, 77).astype(np.int32),
                                                                         np.arange(12, dtype=np.int64).reshape(3, 4),
                                                                         (rng.random(9) > .5), np.zeros(0),
                                                                         np.array("2026-05-01T00:00:03", "datetime64[ns]").reshape(1)]
        for _ in range(4):
            busy.fill_(float(rep))
        ups = [ops.to_device(a) for a in arrs]
        doubled = [(u * 2) if u.dtype.is_floating_point else u.clone() for u in ups]  # launched AFTER the uploads
        for a, u, d2 in zip(arrs, ups, doubled):
            exp = a.view(np.int64) if a.dtype.kind == "M" else a
            assert tuple(u.shape) == a.shape
            np.testing.assert_array_equal(u.cpu().numpy(), exp)
            np.testing.assert_array_equal(d2.cpu().numpy(), exp * 2 if exp.dtype.kind == "f" else exp)
    up = ops._uploaders[torch.device("cuda", torch.cuda.current_device())]
    assert 1 <= len(up.pool) <= up._MAX_POOL
    n_pool = len(up.pool)
    for _ in range(50):
        ops.to_device(rng.standard_normal(1000))
    torch.cuda.synchronize()
    assert len(up.pool) <= n_pool + 50 and len(up.pool) <= up._MAX_POOL
    # float32 target dtype, a read-only broadcast view, a torch CPU tensor
    np.testing.assert_array_equal(ops.to_device(np.arange(5.0), dtype=torch.float32).cpu().numpy(), np.arange(5, dtype=np.float32))
    np.testing.assert_array_equal(ops.to_device(np.broadcast_to(np.arange(3.0), (2, 3))).cpu().numpy(), np.tile(np.arange(3.0), (2, 1)))
    np.testing.assert_array_equal(ops.to_device(torch.arange(4)).cpu().numpy(), np.arange(4))
    # fetch_async: the value at the point of the call, whatever is queued afterwards
    t = torch.arange(3, dtype=torch.float64, device="cuda") + 1
    fut = ops.fetch_async(t)
    t.mul_(100)                      # queued after the fetch: must not be seen
    for _ in range(4):
        busy.fill_(1.0)
    assert fut.tolist() == [1.0, 2.0, 3.0] and fut.cpu().tolist() == [1.0, 2.0, 3.0]
    assert ops.fetch_async(t[:1]).item() == 100.0
    # futures nobody reads give their staging buffers back: the pinned pool does not grow with the number of calls
    dl = ops._downloaders[t.device]
    for _ in range(200):
        ops.fetch_async(t)
    assert len(dl.pool) <= 4, len(dl.pool)
    assert ops.fetch_async(t).tolist() == [100.0, 200.0, 300.0]


def _resident_case(ep, C=3, P=400, S=1024, seed=3):
    d = ep.synth.ek60_numpy(C, P, S, seed=seed, ss_every=7)
    d["backscatter_r"][1, 5, S - 60:] = np.nan
    return d, ep.echodata.from_ek60_arrays(d).to_device()


def test_two_calls_on_resident_echodata_make_no_host_synchronisation(env, monkeypatch):
    """torch's synchronisation debug mode raises on any blocking call (tensor.cpu(), .item(), a pageable upload):
    compute_Sv + compute_MVBS on device-resident echodata make none; reading the result does."""
    torch, ep = env
    from echopype_amd.xr_lite import DeferredDataset

    _, ed = _resident_case(ep)
    logging.disable(logging.WARNING)
    try:
        for _ in range(2):  # warm-up: the staging pools, the allocator's blocks
            ep.commongrid.compute_MVBS(ep.calibrate.compute_Sv(ed), range_bin="1m", ping_time_bin="20s")["Sv"].shape
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            ds = ep.calibrate.compute_Sv(ed)
            mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
            ds2 = ep.calibrate.compute_Sv(ed)  # the next file is launched before anything is read
            mv2 = ep.commongrid.compute_MVBS(ds2, range_bin="1m", ping_time_bin="20s")
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert isinstance(mv, DeferredDataset) and not mv.resolved and not mv2.resolved
        shape = mv["Sv"].shape
        assert mv.resolved and not mv2.resolved and mv2["Sv"].shape == shape
        # ... and with EPA_DEFER_MVBS=0 the call assembles before it returns
        monkeypatch.setenv("EPA_DEFER_MVBS", "0")
        mv3 = ep.commongrid.compute_MVBS(ep.calibrate.compute_Sv(ed), range_bin="1m", ping_time_bin="20s")
        assert not isinstance(mv3, DeferredDataset)
        # (a bin's samples are added by floating-point LDS atomics in whatever order the lanes arrive: two runs of the
        #  same kernel agree to rounding, not bit for bit)
        np.testing.assert_allclose(mv3["Sv"].values, mv["Sv"].values, rtol=1e-12, atol=1e-12, equal_nan=True)
        assert mv3["Sv"].attrs == mv["Sv"].attrs and mv3.attrs.keys() == mv.attrs.keys()
    finally:
        logging.disable(logging.NOTSET)


@pytest.mark.parametrize("range_var_max", [None, "150m"])
def test_deferred_result_equals_the_eager_route_and_the_oracle(env, monkeypatch, range_var_max):
    """Same MVBS (values, coordinates, attributes) as with EPA_DEFER_SV=0 (K1, then the binning kernel on the Sv array,
    everything read back inside the call), and as the oracle; the host-side range bound only sizes the launch."""
    torch, ep = env
    from oracle import calibrate as ocal
    from oracle import commongrid as ogrid

    d, ed = _resident_case(ep, seed=11)
    logging.disable(logging.WARNING)
    try:
        kw = dict(range_bin="2m", ping_time_bin="10s", range_var_max=range_var_max)
        ds = ep.calibrate.compute_Sv(ed)
        assert ds["Sv"].data.source.reach_bound is not None  # (host mirrors of sample_interval / sound_speed)
        mv = ep.commongrid.compute_MVBS(ds, **kw)
        monkeypatch.setenv("EPA_DEFER_SV", "0")
        ds_e = ep.calibrate.compute_Sv(ed)
        mv_e = ep.commongrid.compute_MVBS(ds_e, **kw)
    finally:
        logging.disable(logging.NOTSET)
    for k in ("echo_range", "ping_time", "channel"):
        np.testing.assert_array_equal(mv[k].values, mv_e[k].values)
    got, exp = mv["Sv"].values, mv_e["Sv"].values
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-12, equal_nan=True)
    assert mv["Sv"].attrs == mv_e["Sv"].attrs
    np.testing.assert_array_equal(ds["Sv"].values, ds_e["Sv"].values)
    gain = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["gain_correction"])
    sa = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["sa_correction"])
    sv, er = ocal.cal_power_ek(
        d["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=d["sample_interval"],
        sound_speed=d["sound_speed_indicative"], absorption=d["absorption_indicative"],
        transmit_power=d["transmit_power"], tau_nominal=d["transmit_duration_nominal"], gain=gain, sa_correction=sa,
        psi=d["equivalent_beam_angle"], f_nominal=d["frequency_nominal"], tau_eff=d["transmit_duration_nominal"][:, 0])
    exp_mv, _, r_left = ogrid.compute_MVBS(sv, er, d["ping_time"], "2m", "10s", range_var_max=range_var_max)
    np.testing.assert_array_equal(mv["echo_range"].values, r_left)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp_mv))
    f = np.isfinite(exp_mv)
    assert np.max(np.abs(got[f] - exp_mv[f]) / np.maximum(np.abs(exp_mv[f]), 1.0)) < 1e-9


def test_deferred_error_surfaces_at_first_use(env):
    """An all-NaN file: the reference's compute_MVBS raises on the empty range grid; here the same error comes out when
    the dataset is first touched (the kernel that finds out has only been launched when the call returns)."""
    torch, ep = env
    d = ep.synth.ek60_numpy(2, 60, 512)
    d["backscatter_r"][:] = np.nan
    ed = ep.echodata.from_ek60_arrays(d).to_device()
    logging.disable(logging.WARNING)
    try:
        ds = ep.calibrate.compute_Sv(ed)
        mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
        with pytest.raises(ValueError, match="range bins are empty"):
            mv["Sv"]
        # ... and again at every later access: the SAME error, not a generic "failed earlier"
        with pytest.raises(ValueError, match="range bins are empty"):
            mv.attrs
    finally:
        logging.disable(logging.NOTSET)


def test_deferred_result_does_not_follow_later_edits_of_the_input_dataset(env):
    """What the deferred assembly needs from ds_Sv is taken when compute_MVBS is CALLED: replacing variables or editing
    attributes of the input afterwards does not change the result (the reference's result never depends on what happens
    to its input after the call)."""
    torch, ep = env
    d, ed = _resident_case(ep)
    logging.disable(logging.WARNING)
    try:
        ds = ep.calibrate.compute_Sv(ed)
        mv_ref = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
        ref_vals, ref_chan = mv_ref["Sv"].values, list(mv_ref["channel"].values)
        ref_freq = mv_ref["frequency_nominal"].values.copy()
        ds2 = ep.calibrate.compute_Sv(ed)
        mv = ep.commongrid.compute_MVBS(ds2, range_bin="1m", ping_time_bin="20s")
        assert not mv.resolved
        ds2["frequency_nominal"] = (("channel",), np.zeros(len(ref_chan)))   # edits AFTER the call
        ds2["Sv"] = (("channel", "ping_time", "range_sample"), np.zeros(ds2["Sv"].shape))
        ds2.attrs["processing_level"] = "edited"
        # (two runs of the kernel add a bin's partial sums in LDS in whatever order the wavefronts arrive: an ulp)
        np.testing.assert_allclose(mv["Sv"].values, ref_vals, rtol=1e-13, atol=0, equal_nan=True)
        np.testing.assert_array_equal(mv["frequency_nominal"].values, ref_freq)
        assert mv.attrs.get("processing_level") == mv_ref.attrs.get("processing_level")
    finally:
        logging.disable(logging.NOTSET)


# ---- the chain compute_Sv -> remove_background_noise -> compute_MVBS as three reference calls -------------------------
def _three_calls(ep, ed, **mv_kw):
    ds = ep.calibrate.compute_Sv(ed)
    ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50)
    corrected = ds.copy()
    corrected["Sv"] = ds["Sv_corrected"]
    return ds, ep.commongrid.compute_MVBS(corrected, **mv_kw)


def test_three_calls_of_the_chain_wait_for_nothing_and_run_two_passes(env):
    """compute_Sv leaves Sv deferred; remove_background_noise runs pass 1 (Sv + the noise estimate) and leaves
    Sv_noise / Sv_corrected deferred (their actual_range attributes too: by-products of the kernel that will write
    them); compute_MVBS of the dataset with Sv := Sv_corrected runs pass 2 on its grid: one sweep writes both arrays
    AND the bins.  No host synchronisation anywhere; two kernels over the samples, no separate binning of an array."""
    torch, ep = env
    from echopype_amd import _lib
    from echopype_amd.xr_lite import DeferredDataset, LazyAttrs, LazyDeviceArray

    _, ed = _resident_case(ep)
    kw = dict(range_bin="1m", ping_time_bin="20s")
    logging.disable(logging.WARNING)
    try:
        for _ in range(2):
            _three_calls(ep, ed, **kw)[1]["Sv"].shape
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            with _lib.launch_trace() as tr:
                ds, mv = _three_calls(ep, ed, **kw)
                ds2, mv2 = _three_calls(ep, ed, **kw)  # the next file, before anything of the first is read
        finally:
            torch.cuda.set_sync_debug_mode("default")
        big = [k for k in tr.kernels if k.startswith(("sv_noise", "sv_denoise", "mvbs_of", "block_reduce", "sv_power",
                                                       "fused_sv", "noise_"))]
        # (pass 2 + bins: one of the chain kernels at full size, the generic reduction at this one -- ONE launch either way)
        assert len(big) == 4 and big[0].startswith("sv_noise") and big[2].startswith("sv_noise"), tr.kernels
        assert all(k.startswith(("sv_denoise_mvbs", "block_reduce")) for k in (big[1], big[3])), tr.kernels
        assert isinstance(mv, DeferredDataset) and not mv.resolved
        for name in ("Sv_noise", "Sv_corrected"):
            v = ds.data_vars[name]
            assert isinstance(v.data, LazyDeviceArray) and v.data.materialized          # written by compute_MVBS's pass
            assert isinstance(v.attrs, LazyAttrs) and v.attrs.has_pending("actual_range")
        lo, hi = ds["Sv_corrected"].attrs["actual_range"]
        sc = ds["Sv_corrected"].values
        assert lo == round(float(np.nanmin(sc)), 2) and hi == round(float(np.nanmax(sc)), 2)
        assert list(ds["Sv_noise"].attrs)[:4] == ["long_name", "units", "actual_range", "noise_ping_num"]
        assert mv["Sv"].shape == mv2["Sv"].shape
    finally:
        logging.disable(logging.NOTSET)


@pytest.mark.parametrize("read_first", [False, True])
def test_three_calls_deferred_equal_eager(env, monkeypatch, read_first):
    """Same datasets as with EPA_DEFER_SV=0 / EPA_DEFER_CLEAN=0 (every call writes its arrays before it returns and
    the binning reads the corrected array) -- also when somebody reads Sv_corrected between the calls (pass 2 then runs
    alone and compute_MVBS bins the array)."""
    torch, ep = env
    _, ed = _resident_case(ep, seed=17)
    kw = dict(range_bin="2m", ping_time_bin="10s")
    logging.disable(logging.WARNING)
    try:
        ds = ep.calibrate.compute_Sv(ed)
        ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50, background_noise_max="-125.0dB")
        if read_first:
            first = ds["Sv_corrected"].values.copy()
        corrected = ds.copy()
        corrected["Sv"] = ds["Sv_corrected"]
        mv = ep.commongrid.compute_MVBS(corrected, **kw)
        monkeypatch.setenv("EPA_DEFER_SV", "0")
        monkeypatch.setenv("EPA_DEFER_CLEAN", "0")
        ds_e = ep.calibrate.compute_Sv(ed)
        ep.clean.remove_background_noise(ds_e, ping_num=20, range_sample_num=50, background_noise_max="-125.0dB")
        corrected_e = ds_e.copy()
        corrected_e["Sv"] = ds_e["Sv_corrected"]
        mv_e = ep.commongrid.compute_MVBS(corrected_e, **kw)
    finally:
        logging.disable(logging.NOTSET)
    for name in ("Sv", "Sv_noise", "Sv_corrected"):
        # (the passes over the raw samples and the array kernels contract their multiplications differently: 1e-16)
        np.testing.assert_array_equal(np.isnan(ds[name].values), np.isnan(ds_e[name].values))
        np.testing.assert_allclose(ds[name].values, ds_e[name].values, rtol=1e-13, atol=0, equal_nan=True)
        assert dict(ds[name].attrs) == dict(ds_e[name].attrs), name
    if read_first:
        np.testing.assert_array_equal(first, ds["Sv_corrected"].values)
    for k in ("echo_range", "ping_time", "channel"):
        np.testing.assert_array_equal(mv[k].values, mv_e[k].values)
    np.testing.assert_array_equal(np.isnan(mv["Sv"].values), np.isnan(mv_e["Sv"].values))
    np.testing.assert_allclose(mv["Sv"].values, mv_e["Sv"].values, rtol=1e-12, atol=1e-12, equal_nan=True)
    assert dict(mv["Sv"].attrs) == dict(mv_e["Sv"].attrs) and set(ds.attrs) == set(ds_e.attrs)


def test_deferred_outputs_dropped_unread_free_their_inputs_at_once(env):
    """The deferred Sv_noise / Sv_corrected own their DenoiseSource (which holds the raw samples, the noise estimate and
    the Sv of pass 1), not the other way round: a dataset dropped before anybody read them is freed by reference
    counting -- no cycle for the collector to find while 30-GB arrays wait."""
    import gc
    import weakref

    torch, ep = env
    _, ed = _resident_case(ep, seed=5)
    logging.disable(logging.WARNING)
    try:
        gc.collect()
        gc.disable()
        try:
            ds = ep.calibrate.compute_Sv(ed)
            ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50)
            lazy = ds.data_vars["Sv_corrected"].data
            refs = [weakref.ref(lazy), weakref.ref(lazy.source), weakref.ref(ds.data_vars["Sv"].data)]
            assert not lazy.materialized
            del lazy, ds
            assert [r() for r in refs] == [None, None, None]
        finally:
            gc.enable()
    finally:
        logging.disable(logging.NOTSET)


def _ek80_resident(ep, C=2, P=300, S=2048, B=4, seed=2):
    import torch

    d = ep.synth.ek80_numpy(C, 4, 64, B)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    re = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
    im = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
    re[1, 7, S - 90:] = float("nan")
    p = np.arange(P)
    d.update(backscatter_r=ep.DeviceArray(re), backscatter_i=ep.DeviceArray(im), sample_interval=np.full((C, P), 8e-6),
             sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e3), (C, 1)),
             ping_time=np.datetime64("2026-05-01T00:00:00", "ns") + (p * 1_000_000_000).astype("timedelta64[ns]"))
    return ep.echodata.from_ek80_arrays(d, ep.synth.ek80_filters()).to_device()


def test_ek80_broadband_two_calls_wait_for_nothing_and_equal_the_eager_route(env, monkeypatch):
    """EK80 BB complex samples: compute_Sv runs its pulse-compression kernel (echo_range stays lazy: coefficient rows +
    the kernel's {nanmin, nanmax, NaN count}); compute_MVBS bins the Sv ARRAY on the grid the range's host-side bound
    gives and returns a DeferredDataset trimmed on first use -- no host synchronisation in either call (the gain table
    is looked up on the host mirror of the pulse lengths).  Same dataset as with EPA_DEFER_MVBS=0."""
    torch, ep = env
    from echopype_amd.xr_lite import DeferredDataset

    ed = _ek80_resident(ep)
    kw = dict(waveform_mode="BB", encode_mode="complex")
    mkw = dict(range_bin="0.5m", ping_time_bin="20s")
    logging.disable(logging.WARNING)
    try:
        for _ in range(2):
            ep.commongrid.compute_MVBS(ep.calibrate.compute_Sv(ed, **kw), **mkw)["Sv"].shape
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            ds = ep.calibrate.compute_Sv(ed, **kw)
            mv = ep.commongrid.compute_MVBS(ds, **mkw)
            mv_b = ep.commongrid.compute_MVBS(ep.calibrate.compute_Sv(ed, **kw), **mkw)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert isinstance(mv, DeferredDataset) and not mv.resolved and ds["echo_range"].data.reach_bound > 0
        monkeypatch.setenv("EPA_DEFER_MVBS", "0")
        mv_e = ep.commongrid.compute_MVBS(ep.calibrate.compute_Sv(ed, **kw), **mkw)
        assert not isinstance(mv_e, DeferredDataset)
    finally:
        logging.disable(logging.NOTSET)
    for k in ("echo_range", "ping_time", "channel"):
        np.testing.assert_array_equal(mv[k].values, mv_e[k].values)
    np.testing.assert_array_equal(np.isnan(mv["Sv"].values), np.isnan(mv_e["Sv"].values))
    np.testing.assert_allclose(mv["Sv"].values, mv_e["Sv"].values, rtol=1e-12, atol=1e-12, equal_nan=True)
    assert mv["Sv"].shape == mv_b["Sv"].shape and dict(mv["Sv"].attrs) == dict(mv_e["Sv"].attrs)
    assert mv["echo_range"].values[-1] <= float(np.nanmax(ds["echo_range"].values)) < ds["echo_range"].data.reach_bound


@pytest.mark.gpu
def test_ping_time_is_kept_only_for_resident_data_and_small_uploads_by_content():
    """ops.ping_time_facts keeps the int64 twin / sortedness of a ping_time array with the ARRAY OBJECT only when nobody
    can write to it (EchoData.to_device marks it so); a plain host dataset whose ping_time is edited in place between two
    calls is looked at again.  ops.to_device_small hands out one device copy per distinct content."""
    import torch

    import echopype_amd as ep
    from echopype_amd import ops

    d = ep.synth.ek60_numpy(2, 120, 600, ss_every=1000)
    ed = ep.echodata.from_ek60_arrays(d)
    ds = ep.calibrate.compute_Sv(ed)
    a = ep.commongrid.compute_MVBS(ds, range_bin="5m", ping_time_bin="20s")
    n_a = a["Sv"].shape[1]
    pt = ds["ping_time"].values
    assert pt.flags.writeable
    pt += np.timedelta64(1000, "s") * np.arange(pt.size)  # in place: the pings now spread over many more bins
    ds2 = ep.calibrate.compute_Sv(ed)
    b = ep.commongrid.compute_MVBS(ds2, range_bin="5m", ping_time_bin="20s")
    assert b["Sv"].shape[1] > 10 * n_a
    np.testing.assert_array_equal(b["ping_time"].values[0], a["ping_time"].values[0])
    # resident: read-only, looked at once
    ed_r = ep.echodata.from_ek60_arrays(ep.synth.ek60_numpy(2, 120, 600, ss_every=1000)).to_device()
    ptr = ed_r["Sonar/Beam_group1"]["ping_time"].values
    assert not ptr.flags.writeable
    with pytest.raises(ValueError):
        ptr[0] = ptr[1]
    ns1, ok1, t1 = ops.ping_time_facts(ptr)
    ns2, ok2, t2 = ops.ping_time_facts(ptr)
    assert ok1 and ok2 and t1 is t2 and ns1 is ns2
    np.testing.assert_array_equal(t1.cpu().numpy(), ptr.view(np.int64))
    # small uploads: one tensor per content, another for other content; large arrays are not kept
    x = np.arange(5, dtype=np.float64)
    u, v, w = ops.to_device_small(x), ops.to_device_small(x.copy()), ops.to_device_small(x + 1)
    assert u is v and w is not u and torch.equal(w, u + 1)
    big = np.zeros(4096)
    assert ops.to_device_small(big) is not ops.to_device_small(big)
