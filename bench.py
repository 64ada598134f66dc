#!/usr/bin/env python3
"""bench.py -- range-samples/s through compute_Sv -> compute_MVBS on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json): synthetic EK60 CW, 4 channels x 500 000 pings x 2000 range samples PER
GPU (configs[1]'s volume, 4.0 G samples, through the metric's path compute_Sv -> compute_MVBS with
20-s x 1-m bins, fp64), generated in HBM; pings shard across ranks (weak scaling: the 8-GPU run is
the 2 M-ping job of configs[4] at the range depth of configs[1]).
One STEP = the whole hot path over the resident volume:
    epa_power_coef_ek (K0)  ->  epa_time_bin_offsets  ->  epa_sv_mvbs_fused (K1+K5: writes Sv f64 and
    the MVBS grid)  [-> straddling-bin all-reduce when a time bin crosses a shard edge; not the case
    for this layout, so the data path has no collective].
value = samples processed by all ranks / max-over-ranks wall time of the K timed steps.
roofline: HIP-event time of the dominant kernel (epa_sv_mvbs_fused) on torch's stream, algorithmic
bytes = 12 B/sample (4 B f32 raw in + 8 B f64 Sv out; SURVEY 8d line C) vs 8 TB/s HBM peak.
cpu_baseline: the NumPy oracle (reference pass structure) on a bounded slice, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (C, pings per GPU, S)
    "cfg2": (4, 500_000, 2000),
    # BASELINE configs[2]: the same volume through the WHOLE chain compute_Sv -> remove_background_noise ->
    # compute_MVBS(Sv_corrected) (two sweeps of the raw power, Sv and Sv_corrected written); not the headline
    "cfg3": (4, 500_000, 2000),
    "cfg5shard": (4, 250_000, 4096),
    # BASELINE configs[4] in full: 4 x 2 M x 4096 (32.8 G samples, 131 GB raw, 262 GB of Sv) split
    # over the ranks (STRONG scaling) and, per rank, into resident 250 k-ping tiles ("files") whose Sv
    # goes to one reused 32.8 GB buffer -- the volume does not fit 288 GB otherwise
    "cfg5": (4, 2_000_000, 4096),
    # BASELINE configs[3]: EK80 broadband, 2 ch x 200 000 pings x 8192 samples x 4 sectors per GPU, complex samples as
    # float32 planes (105 GB resident): pulse compression + Sv (K3+K4) then MVBS (20 s x 0.1 m); own run function
    "cfg4": (2, 200_000, 8192),
    "cfg4small": (2, 20_000, 8192),
    "small": (4, 20_000, 2000),
    "straddle": (4, 20_010, 2000),  # shard edges cut a time bin: exercises the edge-bin all-reduce
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SAMPLE = {"float64": 12, "float32": 8}  # SURVEY 8d line C (raw f32 in + Sv out)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--input", default="float32", choices=["float32", "int16"],
                    help="float32: backscatter_r as echopype's converter stores it (the drop-in boundary); "
                         "int16: the instrument's own samples + ping lengths (SURVEY 8f row 4, 2 B/sample in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain-outputs", default="all", choices=["all", "corrected"],
                    help="cfg3: full-size arrays written -- all = Sv + Sv_noise + Sv_corrected (what "
                         "remove_background_noise adds, SURVEY 8d line E: 32 B/sample fp64); corrected = without Sv_noise")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL; gloo only for dry runs of the N>1 logic)")
    ap.add_argument("--single-device", action="store_true",
                    help="dry run: every rank uses cuda:0 (with --backend gloo)")
    return ap.parse_args()


def cpu_baseline(dtype):
    """Oracle chain (reference pass structure, NumPy fp64, one core) on a 4 x 2500 x 2000 slice of
    the same synthetic recipe: compute_Sv then compute_MVBS."""
    from oracle import calibrate as ocal
    from oracle import commongrid as ogrid
    from echopype_amd import synth

    C, P, S = 4, 2500, 2000
    d = synth.ek60_numpy(C, P, S)

    def run():
        gain = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["gain_correction"])
        sa = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["sa_correction"])
        sv, er = ocal.cal_power_ek(
            d["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=d["sample_interval"],
            sound_speed=d["sound_speed_indicative"], absorption=d["absorption_indicative"],
            transmit_power=d["transmit_power"], tau_nominal=d["transmit_duration_nominal"], gain=gain,
            sa_correction=sa, psi=d["equivalent_beam_angle"], f_nominal=d["frequency_nominal"],
            tau_eff=d["transmit_duration_nominal"][:, 0])
        return ogrid.compute_MVBS(sv, er, d["ping_time"], "1m", "20s")

    run()
    ts = []
    t_end = time.perf_counter() + 20.0
    while len(ts) < 5 and (not ts or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    n = C * P * S
    out = {"value": n / float(np.median(ts)), "unit": "range-samples/s", "cores": 1, "kind": "port",
           "sample": f"EK60 {C}ch x {P} pings x {S} range, compute_Sv + compute_MVBS(20s x 1m), NumPy fp64 "
                     f"oracle (reference pass structure), median of {len(ts)} runs, host has {os.cpu_count()} cores"}
    # what dask chunk-parallelism over ping_time could reach at best: the same slice in N processes
    try:
        import multiprocessing as mp

        ncore = max(1, min(32, (os.cpu_count() or 1) // 2))
        if ncore > 1:
            with mp.get_context("fork").Pool(ncore) as pool:
                t0 = time.perf_counter()
                pool.map(_cpu_worker, [(C, P, S, i) for i in range(ncore)])
                dtm = time.perf_counter() - t0
            out["multicore"] = {"value": n * ncore / dtm, "cores": ncore,
                                "sample": f"{ncore} processes x the same slice (ping-sharded, no communication)"}
    except Exception as e:  # noqa: BLE001 - the single-core figure stands on its own
        out["multicore"] = {"error": repr(e)}
    return out


def _cpu_worker(a):
    C, P, S, seed = a
    from oracle import calibrate as ocal
    from oracle import commongrid as ogrid
    from echopype_amd import synth

    d = synth.ek60_numpy(C, P, S, seed=20260501 + seed)
    gain = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["gain_correction"])
    sa = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["sa_correction"])
    sv, er = ocal.cal_power_ek(
        d["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=d["sample_interval"],
        sound_speed=d["sound_speed_indicative"], absorption=d["absorption_indicative"],
        transmit_power=d["transmit_power"], tau_nominal=d["transmit_duration_nominal"], gain=gain,
        sa_correction=sa, psi=d["equivalent_beam_angle"], f_nominal=d["frequency_nominal"],
        tau_eff=d["transmit_duration_nominal"][:, 0])
    ogrid.compute_MVBS(sv, er, d["ping_time"], "1m", "20s")
    return 0


def run_ek80(args, torch, dist, ops, sharding, synth, world, rank, C, P, S, dt):
    """cfg4: EK80 BB complex -> pulse compression + Sv (epa_sv_complex_fft) -> MVBS.  The per-(channel, ping)
    parameter rows and the replicas are assembled once by the drop-in's own calibrator (host, O(C*P)); one step =
    the two kernels over the resident planes."""
    import echopype_amd as ep
    from echopype_amd import _lib
    from echopype_amd.calibrate.api import CALIBRATOR

    B = 4
    d = synth.ek80_numpy(C, 4, 64, B)  # parameters only; the sample planes are generated on the device
    g = torch.Generator(device="cuda")
    g.manual_seed(20260504 + rank)
    re = torch.empty((C, P, S, B), dtype=torch.float32, device="cuda")
    im = torch.empty((C, P, S, B), dtype=torch.float32, device="cuda")
    slab = max(1, P // 50)
    for p0 in range(0, P, slab):  # in slabs: no second copy of the planes
        n = min(slab, P - p0)
        re[:, p0:p0 + n] = torch.randn((C, n, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
        im[:, p0:p0 + n] = torch.randn((C, n, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
    nan_pings = torch.rand(P, generator=g, device="cuda") < 0.10
    tail = int(round(0.05 * S))
    re[:, nan_pings, S - tail:] = float("nan")
    im[:, nan_pings, S - tail:] = float("nan")
    pidx = np.arange(P) + rank * P
    ping_time = np.datetime64("2026-05-01T00:00:00", "ns") + (pidx * 1_000_000_000).astype("timedelta64[ns]")
    d.update(backscatter_r=ep.DeviceArray(re), backscatter_i=ep.DeviceArray(im), sample_interval=np.full((C, P), 8e-6),
             sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * pidx / 1e5), (C, 1)), ping_time=ping_time)
    ed = ep.echodata.from_ek80_arrays(d, synth.ek80_filters())
    cal = CALIBRATOR["EK80"](ed, env_params=None, cal_params=None, ecs_file=None, waveform_mode="BB",
                             encode_mode="complex", dtype=args.dtype, device=None)
    k, _ = cal._complex_inputs("Sv")
    # time bins (20 s) and the range grid of compute_MVBS; echo_range = s * sample_interval * sound_speed / 2
    ns = torch.from_numpy(ping_time.astype(np.int64)).cuda()
    bin_ns = 20_000_000_000
    e0, _ = sharding.global_time_grid(ns.cpu().numpy(), bin_ns)
    first_bin, last_bin = sharding.local_bin_span(ns.cpu().numpy(), e0, bin_ns)
    n_t = last_bin - first_bin + 1
    range_bin = 0.1
    rmax = sharding.global_max(float((S - 1) * 8e-6 * 1500.5 / 2))
    n_r = len(np.arange(0, rmax + range_bin, range_bin)) - 1
    rows = torch.zeros((C, P, _lib.NCOEF), dtype=torch.float64, device="cuda")
    rows[..., _lib.CF_RA] = k["ccoef"][..., _lib.CC_RA]
    rows[..., _lib.CF_RB] = k["ccoef"][..., _lib.CC_RB]
    n_out = C * P * S
    timers = [ops.Timer() for _ in range(args.steps)]

    def step(timer=None):
        bs = ops.time_bin_offsets(ns, e0 + first_bin * bin_ns, bin_ns, n_t)
        if timer is not None:
            timer.start()
        res = ops.sv_complex(k["re"], k["im"], k["ccoef"], replica=k["replica"], replica_off=k["replica_off"],
                             max_taps=k["max_taps"], dtype=dt, want_range=False)
        if timer is not None:
            timer.stop()
        return ops.mvbs(res["out"], bs, n_t, range_bin, n_r, coef=rows)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(timers[i])
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([tm.elapsed_ms() for tm in timers]))
    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        t = t.to(sharding._comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        bps = B * 8 + (8 if args.dtype == "float64" else 4)  # SURVEY 8d line F: complex64 sectors in, Sv out
        achieved = n_out * bps / (kernel_ms * 1e-3) / 1e9
        taps = int(k["max_taps"])
        print(json.dumps({
            "metric": "range-samples/sec through compute_Sv->compute_MVBS", "value": n_out * world * args.steps / elapsed,
            "unit": "range-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.dtype == "float64" else "f32", "data": "synthetic",
            "config": {"workload": f"EK80 BB complex {C}ch x {P} pings x {S} samples x {B} sectors per GPU ({args.workload}), "
                                   f"float32 planes resident, {taps}-tap replica: pulse compression + Sv, then MVBS "
                                   f"(20 s x {range_bin} m); a sample = one (channel, ping, range_sample) output",
                       "pings_total": P * world, "sharding": f"ping_time x{world}",
                       "collective": "none (shard edges on bin edges)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "sv_complex_fft_kernel",
                         "kernel_ms": kernel_ms, "bytes_per_sample": bps,
                         "direct_form_tflops": 8.0 * taps * n_out / (kernel_ms * 1e-3) / 1e12,
                         "note": "with float32 planes this kernel is held by its LDS-resident fp64 FFT (3 workgroups per "
                                 "CU), not by HBM; fed float64 planes (72 B/sample) it moves 3.9-4.1 TB/s"},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_tiled(args, torch, dist, ops, sharding, synth, world, rank, C, P_total, S, dt, cpu=None):
    """cfg5: this rank's share of the 2 M pings as resident tiles of 250 k pings; one step = K0 +
    fused kernel over every tile; Sv of each tile overwrites one reused buffer."""
    tile_p = 250_000
    n_tiles_total = P_total // tile_p
    my_tiles = [t for t in range(n_tiles_total) if t % world == rank] if world <= n_tiles_total else []
    tiles = [synth.ek60_device(C, tile_p, S, seed=20260505 + t) for t in my_tiles]
    bin_ns, n_t = 20_000_000_000, tile_p // 20
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
    sv = torch.empty((C, tile_p, S), dtype=dt, device="cuda")
    mv = [torch.empty((C, n_t, n_r), dtype=dt, device="cuda") for _ in tiles]
    timers = [ops.Timer() for _ in range(args.steps * max(1, len(tiles)))]

    def step(k=None):
        for i, d in enumerate(tiles):
            coef = ops.power_coef_ek(
                d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
                d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
                d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
                pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
            bs = ops.time_bin_offsets(d["ping_time_ns"], int(d["ping_time"][0].astype(np.int64)), bin_ns, n_t)
            tm = timers[k * len(tiles) + i] if k is not None else None
            if tm:
                tm.start()
            ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv[i])
            if tm:
                tm.stop()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([tm.elapsed_ms() for tm in timers])) if tiles else float("nan")
    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        t = t.to(sharding._comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        bps = BYTES_PER_SAMPLE[args.dtype]
        achieved = C * tile_p * S * bps / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "range-samples/sec through compute_Sv->compute_MVBS",
            "value": C * P_total * S * args.steps / elapsed, "unit": "range-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.dtype == "float64" else "f32", "data": "synthetic",
            "config": {"workload": f"EK60 CW {C}ch x {P_total} pings x {S} range TOTAL (cfg5), {n_tiles_total} resident "
                                   f"tiles of {tile_p} pings dealt to {world} rank(s), fused compute_Sv -> compute_MVBS "
                                   "(20 s x 1 m), Sv written to one reused tile buffer + MVBS kept",
                       "sharding": f"ping tiles x{world}", "collective": "none (tile edges on bin edges)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "epa_fused::fused_sv_mvbs_kernel", "kernel_ms": kernel_ms, "bytes_per_sample": bps},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                     f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
    if args.single_device:
        local_rank = 0
    # CPU baseline first: it forks worker processes, which must happen before HIP is initialised
    cpu = cpu_baseline(args.dtype) if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from echopype_amd import ops, sharding, synth

    C, P, S = WORKLOADS[args.workload]
    dt = torch.float64 if args.dtype == "float64" else torch.float32
    if args.workload == "cfg5":
        return run_tiled(args, torch, dist, ops, sharding, synth, world, rank, C, P, S, dt, cpu)
    if args.workload.startswith("cfg4"):
        return run_ek80(args, torch, dist, ops, sharding, synth, world, rank, C, P, S, dt)
    i16 = args.input == "int16"
    d = (synth.ek60_device_i16 if i16 else synth.ek60_device)(C, P, S, seed=20260501 + rank)
    # ping times of this shard: global ping index offset by rank (1 ping / s)
    ns_local = d["ping_time_ns"] + rank * P * 1_000_000_000
    bin_ns = 20_000_000_000
    e0, n_t_global = sharding.global_time_grid(ns_local.cpu().numpy(), bin_ns)
    first_bin, last_bin = sharding.local_bin_span(ns_local.cpu().numpy(), e0, bin_ns)
    n_t = last_bin - first_bin + 1
    e0_local = e0 + first_bin * bin_ns
    # range grid: np.arange(0, max(echo_range) + 1, 1); echo_range max is analytic for this recipe
    r_max = sharding.global_max(float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2))
    n_r = len(np.arange(0, r_max + 1.0, 1.0)) - 1
    straddle = (P % 20) != 0  # shard edges cut a 20-ping bin?

    chain = args.workload == "cfg3"
    if chain and (i16 or straddle):
        sys.exit("cfg3 runs the float32 input on bin-aligned shards")
    sv = torch.empty((C, P, S), dtype=dt, device="cuda") if not chain else None
    mvbs = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
    tau0 = d["transmit_duration_nominal"][:, 0].contiguous()  # EK60 tau_eff = ping 0 (per shard = global here)
    timers = [ops.Timer() for _ in range(args.steps)]

    def step(timer=None):
        coef = ops.power_coef_ek(
            d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
            d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
            d["equivalent_beam_angle"], d["frequency_nominal"], tau0, pulse_length=d["pulse_length"],
            gain_is_table=True, sa_is_table=True)
        bs = ops.time_bin_offsets(ns_local, e0_local, bin_ns, n_t)
        if timer is not None:
            timer.start()
        if chain:  # noise blocks of 20 pings x 50 samples, SNR 3 dB
            a2 = coef[..., 4].contiguous()
            _, _, nz = ops.sv_noise_fused(d["backscatter_r"], coef, a2, 20, 50, dtype=dt)
            res = ops.sv_denoise_mvbs(d["backscatter_r"], coef, a2, nz, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt,
                                      want_noise=args.chain_outputs == "all")
        elif i16:
            res = ops.sv_mvbs_fused_i16(d["raw_i16"], d["n_valid"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv,
                                        mvbs_out=mvbs, want_partials=straddle)
        else:
            res = ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv,
                                    mvbs_out=mvbs, want_partials=straddle)
        if timer is not None:
            timer.stop()
        if straddle and world > 1:
            keep = sharding.merge_straddling_bins(res["sum"], res["cnt"], first_bin, last_bin)
            mvbs.copy_(ops.mvbs_finalize(res["sum"], res["cnt"]))
            del keep

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(timers[i])  # HIP events around the dominant kernel, recorded asynchronously
    sync()
    elapsed = time.perf_counter() - t0
    # average launch duration of the dominant kernel over the timed region (events read after it)
    kernel_ms = float(np.mean([tm.elapsed_ms() for tm in timers]))

    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        t = t.to(sharding._comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    samples_total = C * P * S * world
    value = samples_total * args.steps / elapsed

    if rank == 0:
        bps = BYTES_PER_SAMPLE[args.dtype] - (2 if i16 else 0)
        if chain:
            # two sweeps over the 4-B input + Sv, Sv_corrected (+ Sv_noise) out: SURVEY 8d line E = 32 / 20 B
            bps = 2 * BYTES_PER_SAMPLE[args.dtype] + ((BYTES_PER_SAMPLE[args.dtype] - 4) if args.chain_outputs == "all" else 0)
        achieved = C * P * S * bps / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.workload}:{args.dtype}" + (":int16" if i16 else "") + (
                    ":corrected" if chain and args.chain_outputs != "all" else "")
                if key in tj:
                    traffic = tj[key]["bytes_per_launch"]
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "range-samples/sec through compute_Sv->compute_MVBS" + (
                " with remove_background_noise" if chain else ""),
            "value": value, "unit": "range-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.dtype == "float64" else "f32",
            "data": "synthetic",
            "config": {"workload": f"EK60 CW {C}ch x {P} pings x {S} range per GPU ({args.workload}), "
                                   + ("fused compute_Sv -> compute_MVBS (20 s x 1 m), Sv + MVBS written" if not chain else
                                      "two-pass compute_Sv -> remove_background_noise (20 x 50, 3 dB) -> compute_MVBS of "
                                      "Sv_corrected (20 s x 1 m), Sv + "
                                      + ("Sv_noise + " if args.chain_outputs == "all" else "") + "Sv_corrected + MVBS written")
                                   + (", int16 instrument samples in" if i16 else ""),
                       "pings_total": P * world, "sharding": f"ping_time x{world}",
                       "collective": "none (shard edges on bin edges)" if not straddle else "edge-bin all-reduce"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("epa_chain::sv_noise_fast_kernel + sv_denoise_mvbs_fast_kernel" if chain
                                    else "epa_fused::fused_sv_mvbs_kernel"), "kernel_ms": kernel_ms,
                         "bytes_per_sample": bps},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
