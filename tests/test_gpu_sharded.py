"""The ping-sharded product entry points (echopype_amd.sharding.compute_Sv_MVBS / compute_MVBS /
remove_background_noise) on a real GPU: two and three processes share cuda:0 and talk over gloo (the driver's box has one
GPU; with RCCL the same code runs one rank per GPU), every rank holds a contiguous ping shard whose edges cut MVBS time
bins and noise ping blocks, and the concatenation of the ranks' results must equal the single-process call on the whole
dataset -- which the other GPU tests hold to the oracle and to the reference-executed goldens."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _slice_ek60(d, p0, p1):
    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[1] == d["backscatter_r"].shape[1] and k != "pulse_length" \
                and k not in ("gain_correction", "sa_correction"):
            out[k] = np.ascontiguousarray(v[:, p0:p1])
        elif k == "ping_time":
            out[k] = v[p0:p1]
        else:
            out[k] = v
    return out


def _worker(rank, world, port, split, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import echopype_amd as ep
    from echopype_amd import sharding

    C, P, S = 3, 230, 600
    d = ep.synth.ek60_numpy(C, P, S, seed=77)
    d["ping_time"] = d["ping_time"] + np.timedelta64(7, "s")  # bins of 20 s are cut by every shard edge below
    bounds = [0] + list(split) + [P]
    p0, p1 = bounds[rank], bounds[rank + 1]
    ed = ep.echodata.from_ek60_arrays(_slice_ek60(d, p0, p1))
    tau0 = d["transmit_duration_nominal"][:, 0]
    # (1) fused compute_Sv -> compute_MVBS on the shard
    ds_Sv, mv = sharding.compute_Sv_MVBS(ed, range_bin="2m", ping_time_bin="20s",
                                         tau_effective_first_ping=None if rank == 0 else tau0)
    # (2) compute_Sv, then the sharded noise removal and the sharded MVBS of the corrected Sv
    ds = ep.calibrate.compute_Sv(ed)
    sharding.remove_background_noise(ds, 20, 50, ping_offset=p0, background_noise_max="-100.0dB")
    corrected = ds.copy()
    corrected["Sv"] = ds["Sv_corrected"]
    mv2 = sharding.compute_MVBS(corrected, range_bin="2m", ping_time_bin="20s")
    q.put((rank, p0, p1, np.asarray(mv["Sv"].values), np.asarray(mv["ping_time"].values),
           np.asarray(ds["Sv_noise"].values), np.asarray(ds["Sv_corrected"].values),
           np.asarray(mv2["Sv"].values), np.asarray(mv2["ping_time"].values), np.asarray(mv["echo_range"].values)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", [(113,), (60, 170), (20, 25)])
def test_sharded_entry_points_equal_single_process(split):
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd as ep

    world = len(split) + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, split, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process answers on the whole dataset
    C, P, S = 3, 230, 600
    d = ep.synth.ek60_numpy(C, P, S, seed=77)
    d["ping_time"] = d["ping_time"] + np.timedelta64(7, "s")
    ed = ep.echodata.from_ek60_arrays(d)
    ds_Sv, mv = ep.compute_Sv_MVBS(ed, range_bin="2m", ping_time_bin="20s")
    ds = ep.calibrate.compute_Sv(ed)
    ep.clean.remove_background_noise(ds, 20, 50, background_noise_max="-100.0dB")
    corrected = ds.copy()
    corrected["Sv"] = ds["Sv_corrected"]
    mv2 = ep.commongrid.compute_MVBS(corrected, range_bin="2m", ping_time_bin="20s")

    def cat(i_val, i_time):
        return (np.concatenate([r[i_val] for r in res], axis=1), np.concatenate([r[i_time] for r in res]))

    got, t = cat(3, 4)
    np.testing.assert_array_equal(t, np.asarray(mv["ping_time"].values))  # every bin exactly once, in order
    np.testing.assert_allclose(got, np.asarray(mv["Sv"].values), rtol=1e-12, atol=1e-12, equal_nan=True)
    for r in res:  # the range grid is the whole dataset's on every rank
        np.testing.assert_array_equal(r[9], np.asarray(mv["echo_range"].values))
    sn = np.concatenate([r[5] for r in res], axis=1)
    sc = np.concatenate([r[6] for r in res], axis=1)
    np.testing.assert_allclose(sn, np.asarray(ds["Sv_noise"].values), rtol=1e-12, atol=1e-10, equal_nan=True)
    np.testing.assert_allclose(sc, np.asarray(ds["Sv_corrected"].values), rtol=1e-12, atol=1e-10, equal_nan=True)
    got2, t2 = cat(7, 8)
    np.testing.assert_array_equal(t2, np.asarray(mv2["ping_time"].values))
    np.testing.assert_allclose(got2, np.asarray(mv2["Sv"].values), rtol=1e-12, atol=1e-12, equal_nan=True)
