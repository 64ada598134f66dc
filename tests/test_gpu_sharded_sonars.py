"""Ping shards of the files the EK60 tests do not cover (round-4 review, item 2): EK80 broadband with a channel that
starts recording late (one filter set; ``assume_single_filter_time``; several filter intervals), EK80 CW complex, AZFP
(its reference-executed golden), and the EK60 chain ``compute_Sv -> remove_background_noise -> compute_MVBS`` on the
two-sweep deferred routes.  2 and 3 gloo ranks share cuda:0; every rank calibrates ITS pings with the whole-file scalars
of ``sharding.file_scalars`` (calibrate/api.py:98-197, calibrate_ek.py:113-162, ek80_complex.py:255-282) and bins them
on the whole dataset's grid; concatenated along ping_time the results must be the single-process results on the whole
file (and, where there is one, the golden)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (builder kwargs, calibration kwargs, cuts per world size)
    "ek80bb": (dict(filter_pings=None), dict(waveform_mode="BB", encode_mode="complex"), {2: (17,), 3: (8, 20)}),
    "ek80bb_single": (dict(filter_pings=[0, 12, 25]), dict(waveform_mode="BB", encode_mode="complex",
                                                           assume_single_filter_time=True), {2: (17,), 3: (8, 20)}),
    "ek80bb_intervals": (dict(filter_pings=[0, 12, 25]), dict(waveform_mode="BB", encode_mode="complex"),
                         {2: (17,), 3: (8, 20)}),
    "azfp": (None, {}, {2: (23,), 3: (9, 41)}),
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(name):
    sys.path.insert(0, HERE)
    from shard_cases import azfp_file, ek80_bb_file

    if name == "azfp":
        ed, env = azfp_file()
        return ed, dict(env_params=env)
    return ek80_bb_file(P=40, S=700, **CASES[name][0]), {}


def _outputs(ds, mv):
    out = dict(sv=np.asarray(ds["Sv"].values), er=np.asarray(ds["echo_range"].values),
               mv=np.asarray(mv["Sv"].values), mt=np.asarray(mv["ping_time"].values),
               mr=np.asarray(mv["echo_range"].values))
    if "tau_effective" in ds:
        out["te"] = np.asarray(ds["tau_effective"].values)
    return out


def _worker(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import logging

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, HERE)
    from shard_cases import shard_of

    import echopype_amd as ep
    from echopype_amd import sharding
    from echopype_amd.echodata import BEAM1

    logging.disable(logging.WARNING)
    ed, extra = _build(name)
    P = ed[BEAM1].sizes["ping_time"]
    bounds = [0] + list(CASES[name][2][world]) + [P]
    shard = shard_of(ed, bounds[rank], bounds[rank + 1])
    kw = dict(CASES[name][1], **extra)
    fs = sharding.file_scalars(shard, waveform_mode=kw.get("waveform_mode"), encode_mode=kw.get("encode_mode"))
    ds = sharding.compute_Sv(shard, file_scalars=fs, **kw)
    mv = sharding.compute_MVBS(ds, range_bin="0.5m", ping_time_bin="10s")
    q.put((rank, _outputs(ds, mv)))
    dist.barrier()
    dist.destroy_process_group()


def _close(got, exp, tol, what):
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp), err_msg=what)
    f = np.isfinite(exp)
    assert f.any(), what
    assert np.max(np.abs(got[f] - exp[f])) <= tol, (what, float(np.max(np.abs(got[f] - exp[f]))))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", list(CASES))
def test_sharded_sonar_equals_the_single_process_result(name, world):
    import logging

    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd as ep

    logging.disable(logging.WARNING)
    try:
        ed, extra = _build(name)
        kw = dict(CASES[name][1], **extra)
        ds = ep.calibrate.compute_Sv(ed, **kw)
        mv = ep.commongrid.compute_MVBS(ds, range_bin="0.5m", ping_time_bin="10s")
        exp = _outputs(ds, mv)
    finally:
        logging.disable(logging.NOTSET)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # (the pulse-compression kernel tiles every ping on its own: the same bits whichever rank holds it)
    _close(np.concatenate([o["sv"] for o in res], axis=1), exp["sv"], 1e-9, f"{name} Sv")
    np.testing.assert_array_equal(np.concatenate([o["er"] for o in res], axis=1), exp["er"])
    if "te" in exp:
        for o in res:  # one value per channel, the whole file's -- or the (channel, ping) grid of this shard's pings
            if exp["te"].ndim == 1:
                np.testing.assert_array_equal(o["te"], exp["te"])
    np.testing.assert_array_equal(np.concatenate([o["mt"] for o in res]), exp["mt"])  # every time bin once, in order
    for o in res:
        np.testing.assert_array_equal(o["mr"], exp["mr"])  # the range grid of the whole dataset on every rank
    _close(np.concatenate([o["mv"] for o in res], axis=1), exp["mv"], 1e-9, f"{name} MVBS")
    assert np.isnan(exp["sv"][1, 0]).all() and np.isfinite(exp["sv"][1, -1]).any() if name.startswith("ek80") else True


def test_sharded_azfp_golden():
    """The AZFP golden (reference-executed, tests/golden/ref_chain_goldens.npz) calibrated as two ping shards."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    sys.path.insert(0, HERE)
    from shard_cases import shard_of

    import echopype_amd as ep
    from echopype_amd import sharding

    g = np.load(os.path.join(HERE, "golden", "ref_chain_goldens.npz"))
    C, P, S = g["azfp_counts"].shape
    t = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(2, "s")
    d = dict(backscatter_r=g["azfp_counts"].astype(np.float32), frequency_nominal=np.array([38e3, 125e3, 200e3])[:C],
             channel=[f"ch{i}" for i in range(C)], transmit_duration_nominal=np.tile(g["azfp_tau"][:, None], (1, P)),
             number_of_samples_per_average_bin=g["azfp_N"], digitization_rate=g["azfp_f"], lock_out_index=g["azfp_L"],
             EL=g["azfp_EL"], DS=g["azfp_DS"], TVR=g["azfp_TVR"], VTX0=g["azfp_VTX0"], Sv_offset=g["azfp_Sv_offset"],
             equivalent_beam_angle=g["azfp_equivalent_beam_angle"], temperature=np.full(P, 8.0), ping_time=t)
    ed = ep.echodata.from_azfp_arrays(d)
    cut = P // 2
    parts = []
    for p0, p1 in ((0, cut), (cut, P)):  # (world size 1: the collectives are identities; the shard logic is what runs)
        env = {"salinity": 30.0, "pressure": 50.0,
               "sound_speed": ep.DataArray(np.tile(g["azfp_sound_speed"], (C, 1))[:, p0:p1], ("channel", "ping_time"),
                                           {"channel": d["channel"], "ping_time": t[p0:p1]}),
               "sound_absorption": ep.DataArray(g["azfp_absorption"], ("channel",), {"channel": d["channel"]})}
        for cal, fn in (("Sv", sharding.compute_Sv), ("TS", sharding.compute_TS)):
            ds = fn(shard_of(ed, p0, p1), env_params=env)
            parts.append((cal, ds[cal].values, ds["echo_range"].values))
    for cal in ("Sv", "TS"):
        got = np.concatenate([v for c, v, _ in parts if c == cal], axis=1)
        rng = np.concatenate([r for c, _, r in parts if c == cal], axis=1)
        _close(got, g[f"azfp_{cal}"], 1e-9 * 200, f"AZFP {cal} golden")
        _close(rng, g[f"azfp_echo_range_{cal}"], 1e-11, f"AZFP echo_range {cal} golden")


def _worker_chain(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import logging

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, HERE)
    from test_gpu_multi_rank import _slice_ek60

    import echopype_amd as ep
    from echopype_amd import _lib, sharding

    logging.disable(logging.WARNING)
    C, P, S = 3, 96 * world + 37, 1024
    d = ep.synth.ek60_numpy(C, P, S, seed=78, ss_every=1)
    d["ping_time"] = d["ping_time"] + np.timedelta64(7, "s")
    p0, p1 = sharding.shard_bounds(P, world, rank)
    ed = ep.echodata.from_ek60_arrays(_slice_ek60(d, p0, p1)).to_device()
    shard = sharding.MVBSShard()
    fs = sharding.file_scalars(ed)
    out = {}
    for rep in range(2):  # (the second round finds its exchange plans)
        with _lib.launch_trace() as tr:
            ds = sharding.compute_Sv(ed, file_scalars=fs)
            sharding.remove_background_noise(ds, 20, 50, ping_offset=p0, background_noise_max="-100.0dB", shard=shard)
            corrected = ds.copy()
            corrected["Sv"] = ds["Sv_corrected"]
            mv = sharding.compute_MVBS(corrected, range_bin="2m", ping_time_bin="20s", shard=shard)
            shape = tuple(mv["Sv"].shape)  # (assembles the deferred dataset)
        out[f"kernels{rep}"] = tr.kernels
    out.update(p0=p0, p1=p1, mv=np.asarray(mv["Sv"].values), t=np.asarray(mv["ping_time"].values), shape=shape,
               r=np.asarray(mv["echo_range"].values), sc=np.asarray(ds["Sv_corrected"].values),
               sn=np.asarray(ds["Sv_noise"].values), sv=np.asarray(ds["Sv"].values),
               ar=list(ds["Sv_corrected"].attrs["actual_range"]))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_chain_runs_two_sweeps_and_equals_the_oracle(world):
    """compute_Sv -> remove_background_noise -> compute_MVBS on ping shards: pass 1 (Sv + noise estimate, block phase and
    edge rows for the cross-shard merge) and pass 2 (Sv_noise, Sv_corrected, the bins) are the ONLY sweeps of the samples
    (launch trace), and the ranks' results concatenate to the oracle's on the whole file (clean/api.py:402-430,485-487;
    commongrid/utils.py:614-627)."""
    import torch
    import torch.multiprocessing as mp

    from oracle import clean as oclean
    from oracle import commongrid as ogrid

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    sys.path.insert(0, HERE)
    from test_gpu_multi_rank import TABLES, _oracle_sv

    import echopype_amd as ep

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chain, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    C, P, S = 3, 96 * world + 37, 1024
    d = ep.synth.ek60_numpy(C, P, S, seed=78, ss_every=1)
    d["ping_time"] = d["ping_time"] + np.timedelta64(7, "s")
    sv, er = _oracle_sv(d, {k: d[k] for k in TABLES}, d["transmit_duration_nominal"][:, 0])
    exp_n, exp_c = oclean.remove_background_noise(sv, er, d["absorption_indicative"], 20, 50, "-100.0dB", "3.0dB")
    exp_mv, t_left, r_left = ogrid.compute_MVBS(exp_c, er, d["ping_time"], "2m", "20s")
    for o in res:
        for rep in (0, 1):
            ks = o[f"kernels{rep}"]
            # every kernel that walks the samples: pass 1, then pass 2 + bins (one of the chain kernels at full size --
            # up to three launches of which all but one return at once --, the generic reduction at this size); no K1,
            # no separate estimate / apply / binning of an array
            big = [k for k in ks if k.startswith(("sv_noise", "sv_denoise", "mvbs_of", "block_reduce", "sv_power",
                                                  "fused_sv", "noise_apply", "range_power"))]
            assert big and big[0] == "sv_noise_fast_kernel" and ks.count("sv_noise_fast_kernel") == 1, ";".join(ks)
            rest = big[1:]
            assert rest and (rest == ["block_reduce_kernel"] or all(k.startswith("sv_denoise_mvbs") for k in rest)), ";".join(ks)
            assert "edge_pack_kernel" in ks, ";".join(ks)  # (the cut noise block / time bins went through the exchange)
    _close(np.concatenate([o["sv"] for o in res], axis=1), sv, 1e-9 * 200, "Sv")
    _close(np.concatenate([o["sc"] for o in res], axis=1), exp_c, 1e-9 * 200, "Sv_corrected")
    np.testing.assert_array_equal(np.concatenate([o["t"] for o in res]), t_left)
    for o in res:
        np.testing.assert_array_equal(o["r"], r_left)
        f = np.isfinite(o["sc"])
        assert o["ar"] == [round(float(o["sc"][f].min()), 2), round(float(o["sc"][f].max()), 2)]
    _close(np.concatenate([o["mv"] for o in res], axis=1), exp_mv, 1e-9 * 200, "MVBS of the corrected Sv")


def _worker_vote(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import logging

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, HERE)
    from test_gpu_multi_rank import _slice_ek60

    import echopype_amd as ep
    from echopype_amd import _lib, sharding

    logging.disable(logging.WARNING)
    C, P, S = 2, 150, 1024
    d = ep.synth.ek60_numpy(C, P, S, seed=31, ss_every=1)
    p0, p1 = sharding.shard_bounds(P, world, rank)
    ed = ep.echodata.from_ek60_arrays(_slice_ek60(d, p0, p1)).to_device()
    shard = sharding.MVBSShard()
    fs = sharding.file_scalars(ed)
    out = dict(p0=p0, p1=p1)
    # (a) the two calls; rank 0 has read its Sv before it bins (its Sv is an array, the other rank's still deferred)
    with _lib.launch_trace() as tr:
        ds = sharding.compute_Sv(ed, file_scalars=fs)
        if rank == 0:
            ds["Sv"].values
        mv = sharding.compute_MVBS(ds, range_bin="2m", ping_time_bin="20s", shard=shard)
        out["a_mv"], out["a_t"] = np.asarray(mv["Sv"].values), np.asarray(mv["ping_time"].values)
    out["a_kernels"] = tr.kernels
    # (b) the chain; rank 1 has read its Sv before remove_background_noise, rank 0 its Sv_corrected before compute_MVBS
    ds = sharding.compute_Sv(ed, file_scalars=fs)
    if rank == 1:
        ds["Sv"].values
    sharding.remove_background_noise(ds, 20, 50, ping_offset=p0, shard=shard)
    out["b_sc"] = np.asarray(ds["Sv_corrected"].values)
    # (c) the chain on the deferred routes up to the binning; rank 0 reads Sv_corrected first
    ds = sharding.compute_Sv(ed, file_scalars=fs)
    sharding.remove_background_noise(ds, 20, 50, ping_offset=p0, shard=shard)
    corrected = ds.copy()
    corrected["Sv"] = ds["Sv_corrected"]
    if rank == 0:
        corrected["Sv"].values
    mv = sharding.compute_MVBS(corrected, range_bin="2m", ping_time_bin="20s", shard=shard)
    out["c_mv"], out["c_t"] = np.asarray(mv["Sv"].values), np.asarray(mv["ping_time"].values)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_route_is_voted_on_when_a_rank_cannot_take_it():
    """Whether a rank's Sv is still deferred is rank-local state; the deferred and the plain routes run different
    collectives.  A rank that has read its array must not leave the others waiting in a collective it never joins
    (ADVICE round 5): the route is voted on in the call's first control message, every rank takes the plain route, and
    the results are the whole file's."""
    import torch
    import torch.multiprocessing as mp

    from oracle import clean as oclean
    from oracle import commongrid as ogrid

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    sys.path.insert(0, HERE)
    from test_gpu_multi_rank import TABLES, _oracle_sv

    import echopype_amd as ep

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_vote, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [o for _, o in sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    C, P, S = 2, 150, 1024
    d = ep.synth.ek60_numpy(C, P, S, seed=31, ss_every=1)
    sv, er = _oracle_sv(d, {k: d[k] for k in TABLES}, d["transmit_duration_nominal"][:, 0])
    exp_mv, t_left, _ = ogrid.compute_MVBS(sv, er, d["ping_time"], "2m", "20s")
    _close(np.concatenate([o["a_mv"] for o in res], axis=1), exp_mv, 1e-9 * 200, "MVBS, one rank's Sv read early")
    np.testing.assert_array_equal(np.concatenate([o["a_t"] for o in res]), t_left)
    assert all("fused_sv_mvbs_kernel" not in o["a_kernels"] for o in res)  # (both ranks took the plain route)
    exp_n, exp_c = oclean.remove_background_noise(sv, er, d["absorption_indicative"], 20, 50, None, "3.0dB")
    _close(np.concatenate([o["b_sc"] for o in res], axis=1), exp_c, 1e-9 * 200, "Sv_corrected, one rank's Sv read early")
    exp_mvc, t_left, _ = ogrid.compute_MVBS(exp_c, er, d["ping_time"], "2m", "20s")
    _close(np.concatenate([o["c_mv"] for o in res], axis=1), exp_mvc, 1e-9 * 200, "MVBS, one rank's Sv_corrected read early")
    np.testing.assert_array_equal(np.concatenate([o["c_t"] for o in res]), t_left)
