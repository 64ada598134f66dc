#!/usr/bin/env python3
"""bench.py -- range-samples/s through compute_Sv -> compute_MVBS on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME[:VARIANT]] ...

ONE workload carries the 1 -> 8 GPU curve: BASELINE configs[4], the metric's own volume -- EK60 CW 4 ch x 2 M pings x
4096 range (32.8 G samples) as eight resident tiles ("files") of 250 000 pings, dealt to the N ranks round-robin (STRONG
scaling: N = 1 holds all eight, N = 8 one each), so that tiles j N .. j N + N - 1 are one contiguous dataset sharded by
ping_time over the N ranks.  The HEADLINE (printed LAST at every N: the driver parses the last line) goes through the
PRODUCT ENTRY POINTS, per resident tile: N = 1 the reference's own two calls, calibrate.compute_Sv(echodata) then
commongrid.compute_MVBS(ds_Sv, "1m", "20s"); N > 1 sharding.compute_Sv_MVBS(echodata_shard, shard=MVBSShard()) -- the
same pass on one rank's shard with the cut time bins summed over the ranks (ONE RCCL all-reduce of a few hundred KB per
dataset: epa_edge_pack -> all_reduce -> epa_edge_finalize_mvbs) and nanmax(echo_range) all-reduced in HBM.  The calls
do not wait for the GPU (uploads on a side stream, the MVBS dataset assembled on first use); the bench reads each
result once the NEXT tile has been launched, as a pipeline writing results out would.  The ping-time origin sits 10 s
off the 20-s bin grid, so every tile edge cuts a time bin.  Beside it, from the same invocation: the ops-level harness
(kernels called directly on preallocated buffers, all cut bins of a rank's tiles exchanged) as config.ops_level_ms_per_pass
(its bin-aligned layout is timed as a check and not printed).

A STEP is `passes_per_step` back-to-back passes of the hot path over the resident volume (so that the driver's 20 steps
are a ~2 s region); `value` = samples processed by all ranks in the K timed steps / max-over-ranks wall time.

N = 1 prints, before the headline, one JSON line per single-GPU configuration of BASELINE.json (skipped with
--only-headline or an explicit --workload): cfg3 = configs[2] (compute_Sv -> remove_background_noise -> compute_MVBS;
SURVEY 8d recipe: a new sound speed at every ping; then the every-2000-pings variant and fp32), cfg2 = configs[1] (fp64,
fp32), the reference's own two API calls on the cfg2 volume and its THREE calls of the cfg3 chain (api:chain), cfg4 = configs[3] (EK80 BB pulse compression + Sv, float32
planes -> fp64 / fp32, and float64 planes as the converter stores them).  Every line carries `roofline` (HIP-event
time of the dominant kernel on torch's stream vs the algorithmic bytes of SURVEY 8d) and `cpu_baseline` (the NumPy /
SciPy oracle with the reference's pass structure on a bounded slice, host cores, timed before HIP is initialised).
The headline line also carries the other lines' key figures as flat `also_*` strings.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SAMPLE = {"float64": 12, "float32": 8}  # SURVEY 8d line C (raw f32 in + Sv out)
METRIC = "range-samples/sec through compute_Sv->compute_MVBS"
# name: (C, pings, S, passes per step for f64, for f32)
WORKLOADS = {
    "cfg2": (4, 500_000, 2000, 10, 14),
    "cfg3": (4, 500_000, 2000, 4, 6),
    "api": (4, 500_000, 2000, 5, 8),
    "cfg4": (2, 200_000, 8192, 2, 3),
    "cfg4small": (2, 20_000, 8192, 20, 30),
    "cfg5": (4, 2_000_000, 4096, 2, 3),
    "small": (4, 20_000, 2000, 50, 50),
    "next": (4, 100_000, 2000, 2, 2),  # SURVEY 8f rows through their API entry points (scripts/perf_masks.py's volume)
}
DEFAULT_LINES = ["cfg3", "cfg3:ss2000", "cfg3:f32", "cfg2", "cfg2:f32", "cfg2:int16", "cfg2:int16f32", "cfg2:bins", "cfg2:int16bins", "cfg2:sv", "cfg2:sv32", "api", "api:chain", "api:pcie",
                 "cfg4", "cfg4:f32", "cfg4:planes64", "next:depth", "next:depthw", "next:masks", "next:masks2000", "next:masksidx", "next:nasc", "cfg5:one", "cfg5"]
TILE_PINGS = 250_000
DT = {"f32": "float32", "f64": "float64", "sv32": "float32", "int16f32": "float32"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--passes", type=int, default=None, help="passes over the volume per step (default: per workload)")
    ap.add_argument("--ss-every", type=int, default=1,
                    help="EK60 volumes: the recorded sound speed changes every this many pings (1 = SURVEY 8d's recipe: "
                         "no two pings share a range vector)")
    ap.add_argument("--workload", default=None,
                    help="one line only: cfg2 | cfg3 | cfg4 | cfg5 | api | small | cfg4small, optionally with a variant "
                         "(:f32, :ss2000, :planes64, :int16, :sv, :sv32)")
    ap.add_argument("--only-headline", action="store_true", help="N=1: skip the other single-GPU lines")
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain-outputs", default="all", choices=["all", "corrected"],
                    help="cfg3: full-size arrays written -- all = Sv + Sv_noise + Sv_corrected (SURVEY 8d line E: "
                         "32 B/sample fp64); corrected = without Sv_noise")
    ap.add_argument("--pings-total", type=int, default=None, help="cfg5: total pings (default 2 000 000)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL; gloo only for dry runs of the N>1 logic)")
    ap.add_argument("--single-device", action="store_true", help="dry run: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--sharded-at-1", action="store_true",
                    help="N = 1: a one-rank process group (--backend) and the SHARDED entry points per tile -- the host "
                         "cost of the N > 1 route (control messages, exchange plan, collectives as identities) on one GPU")
    ap.add_argument("--tile-streams", type=int, default=2,
                    help="cfg5: the resident tiles (independent datasets) are dealt round-robin to this many HIP streams -- "
                         "the tail of a tile's kernel runs beside the head of the next tile's")
    ap.add_argument("--tile-pings", type=int, default=None,
                    help="cfg5 (experiments): pings per tile instead of the workload's eight tiles")
    ap.add_argument("--read-lag", type=int, default=1,
                    help="cfg5: the result of a tile is read after this many further tiles have been launched")
    ap.add_argument("--out", default=None, help="also append every JSON line to this file")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------- CPU baselines
def _oracle_ek60(d, chain):
    from oracle import calibrate as ocal
    from oracle import clean as oclean
    from oracle import commongrid as ogrid

    gain = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["gain_correction"])
    sa = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["sa_correction"])
    sv, er = ocal.cal_power_ek(
        d["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=d["sample_interval"],
        sound_speed=d["sound_speed_indicative"], absorption=d["absorption_indicative"],
        transmit_power=d["transmit_power"], tau_nominal=d["transmit_duration_nominal"], gain=gain,
        sa_correction=sa, psi=d["equivalent_beam_angle"], f_nominal=d["frequency_nominal"],
        tau_eff=d["transmit_duration_nominal"][:, 0])
    if chain:
        _, sv = oclean.remove_background_noise(sv, er, d["absorption_indicative"], 20, 50, None, "3.0dB")
    return ogrid.compute_MVBS(sv, er, d["ping_time"], "1m", "20s")


def _time_runs(run, budget_s=12.0, max_runs=5):
    run()
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < max_runs and (not ts or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def _cpu_worker(a):
    """Whole oracle passes over this worker's ping slice until the common deadline (at least one): (passes, busy s)."""
    C, P, S, seed, deadline = a
    from echopype_amd import synth

    d = synth.ek60_numpy(C, P, S, seed=20260501 + seed)
    t0, reps = time.perf_counter(), 0
    while reps == 0 or time.time() < deadline:
        _oracle_ek60(d, False)
        reps += 1
    return reps, time.perf_counter() - t0


def _cpu_quota():
    """The cgroup's CPU limit in cores (None = unlimited / unknown): a box can show 256 hardware threads and grant ten."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 1)
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline_ek60(chain=False, multicore=False):
    """Oracle chain (reference pass structure, NumPy fp64, one core) on BASELINE configs[0]'s shape: EK60 CW
    2 channels x 10 000 pings x 1000 range."""
    from echopype_amd import synth

    C, P, S = 2, 10_000, 1000
    d = synth.ek60_numpy(C, P, S)
    med, n_runs = _time_runs(lambda: _oracle_ek60(d, chain))
    n = C * P * S
    what = "Sv+denoise(20x50,3dB)+MVBS" if chain else "Sv+MVBS"
    out = {"value": n / med, "unit": "range-samples/s", "cores": 1, "kind": "port",
           "sample": f"EK60 {C}x{P}x{S} {what}, NumPy f64 oracle, median of {n_runs}"}
    if multicore:
        # what dask chunk-parallelism over ping_time could reach at best: a ping slice of the same volume in EVERY core
        # the process may use (north_star: "the same box's host cores (core count stated)"), a quarter of the pings per
        # worker (the workers' footprint -- ~0.4 GB each -- stays well inside the host's memory), whole passes until a
        # common 12-s deadline (a fixed number of passes took 160 s of wall on a 256-thread box: the sample is bounded
        # by time, whatever the box grants)
        try:
            import multiprocessing as mp

            ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:
                import psutil

                ncore = max(1, min(ncore, int(psutil.virtual_memory().available // (1 << 30))))  # >= 1 GiB per worker
            except Exception:  # noqa: BLE001
                pass
            if ncore > 1:
                Pw, budget_s = P // 4, 12.0
                with mp.get_context("fork").Pool(ncore) as pool:
                    t0 = time.perf_counter()
                    got = pool.map(_cpu_worker, [(C, Pw, S, i, time.time() + budget_s) for i in range(ncore)], chunksize=1)
                    dtm = time.perf_counter() - t0
                reps = sum(r for r, _ in got)
                out["multicore_value"], out["multicore_cores"] = C * Pw * S * reps / dtm, ncore
                out["multicore_sample"] = (f"{ncore} procs, {reps} x {C}x{Pw}x{S} in {dtm:.0f} s, {os.cpu_count()} hw threads, "
                                           f"cgroup quota {_cpu_quota()}")
        except Exception as e:  # noqa: BLE001 - the single-core figure stands on its own
            out["multicore_error"] = repr(e)[:100]
    return out


def cpu_baseline_bb():
    """EK80 broadband: the SciPy-convolve oracle (the reference's per-(ping, sector) scipy.signal.convolve loop,
    ek80_complex.py:285-313, + sector mean + Sv chain) on a 2 x 500 x 8192 x 4 slice, one core."""
    from oracle import ek80 as oek
    from echopype_amd import synth

    C, P, S, B = 2, 500, 8192, 4
    rng = np.random.default_rng(20260504)
    x = ((rng.standard_normal((C, P, S, B)) + 1j * rng.standard_normal((C, P, S, B))) * 1e-3).astype(np.complex64)
    filt = synth.ek80_filters()
    bb = synth.EK80_BB
    reps = [oek.transmit_replica(1.5e6, bb["tau"][c], 0.05, bb["f_start"][c], bb["f_stop"][c], filt)[0] for c in range(C)]

    def run():
        prx = oek.power_from_complex(x, bb["z_er"][:C, None, None], bb["z_et"][:C, None, None], reps)
        r = np.arange(S)[None, None, :] * 8e-6 * 750.0
        with np.errstate(invalid="ignore", divide="ignore"):
            return 10 * np.log10(prx) + 20 * np.log10(r) + 0.02 * r - 30.0

    med, n_runs = _time_runs(run, budget_s=12.0, max_runs=3)
    return {"value": C * P * S / med, "unit": "range-samples/s", "cores": 1, "kind": "port",
            "sample": f"EK80 BB {C}x{P}x{S}x{B}, {reps[0].size}-tap replicas: scipy.signal.convolve per (ping, sector) + "
                      f"sector mean + Sv, median of {n_runs}, host has {os.cpu_count()} cores"}


# ---------------------------------------------------------------------------------------- helpers
def csrc_hash():
    """Hash of the kernel sources: a measured HBM-traffic figure is only valid for the code it was measured on."""
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "echopype_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h")):
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """(bytes per launch | None, provenance) from profiles/hbm_traffic.json -- PMC counters of an earlier rocprofv3 run
    of this command; dropped when the kernel sources changed since."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        e = json.load(open(path)).get(key)
        if not e:
            return None, "no PMC run of this line"
        if e.get("csrc_sha16") != csrc_hash():
            return None, f"stale: PMC run at csrc {e.get('csrc_sha16')}"
        return e["bytes_per_launch"], f"PMC 2xFETCH+WRITE, csrc {e['csrc_sha16'][:8]}"  # (profiles/hbm_traffic.json names the csv)
    except Exception as ex:  # noqa: BLE001
        return None, f"unreadable profiles/hbm_traffic.json: {ex!r}"[:100]


class Ctx:
    """What every run function needs."""

    def __init__(self, args, world, rank):
        import torch
        import torch.distributed as dist
        from echopype_amd import ops, sharding, synth

        self.args, self.world, self.rank = args, world, rank
        self.torch, self.dist, self.ops, self.sharding, self.synth = torch, dist, ops, sharding, synth
        self.dtype = args.dtype
        self.cache = {}

    @property
    def dt(self):
        return self.torch.float64 if self.dtype == "float64" else self.torch.float32

    def sync(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def passes(self, workload):
        if self.args.passes:
            return self.args.passes
        return WORKLOADS[workload][3 if self.dtype == "float64" else 4]

    def timed(self, one_pass, passes, finish=None, timers_of=None):
        """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks.  A step = ``passes``
        calls of one_pass(timer | None); ``finish()`` (results still in flight are read) runs INSIDE the timed region,
        before the closing synchronize.  Returns (elapsed s, mean HIP-event ms of the regions the passes timed);
        ``timers_of()``: the harness brackets its launches with its own timers (a pipeline that launches ahead of the
        pass it is asked for) and hands them over here."""
        steps, warmup = self.args.steps, self.args.warmup
        timers = [self.ops.Timer() for _ in range(steps * passes)]
        for _ in range(warmup * passes):
            one_pass(None)
        if finish:
            finish()
        self.sync()
        t0 = time.perf_counter()
        for i in range(steps * passes):
            one_pass(timers[i])
        if finish:
            finish()
        self.sync()
        elapsed = time.perf_counter() - t0
        kernel_ms = float(np.mean([tm.elapsed_ms() for tm in (timers_of() if timers_of else timers)]))
        t = self.torch.tensor([elapsed], dtype=self.torch.float64)
        if self.world > 1:
            t = t.to(self.sharding._comm_device())
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item()), kernel_ms

    def free(self):
        import gc

        gc.collect()
        self.torch.cuda.empty_cache()


def line(ctx, *, samples_per_pass, passes, elapsed, scaling, workload, config, roofline, metric=METRIC, cpu=None):
    steps = ctx.args.steps
    out = {"metric": metric, "value": samples_per_pass * passes * steps / elapsed, "unit": "range-samples/s",
           "n_gpus": ctx.world, "steps": steps, "warmup": ctx.args.warmup, "ms_per_step": elapsed / steps * 1e3,
           "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
           "dtype": "f64" if ctx.dtype == "float64" else "f32", "data": "synthetic",
           "config": {"workload": workload, "passes_per_step": passes, "samples_per_step": samples_per_pass * passes,
                      "ms_per_pass": elapsed / steps / passes * 1e3, **config},
           "roofline": roofline}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def roofline(kernel, kernel_ms, bytes_per_launch, bps, traffic_key=None, **extra):
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic, src = measured_traffic(traffic_key) if traffic_key else (None, "not measured for this line")
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_sample": bps,
            **extra}


# ---------------------------------------------------------------------------------------- EK60: cfg2 / cfg3 / api
def _coef(ctx, d):
    return ctx.ops.power_coef_ek(
        d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"], d["sound_speed_indicative"],
        d["absorption_indicative"], d["gain_correction"], d["sa_correction"], d["equivalent_beam_angle"],
        d["frequency_nominal"], d["tau0"], pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)


def ek60_volume(ctx, name, ss_every, i16=False):
    """The cfg2 / cfg3 volume, generated once per shape and kept between lines; the recorded sound speed (O(C P)) is
    rewritten for the ``ss_every`` asked (the samples do not depend on it)."""
    torch = ctx.torch
    C, P, S = WORKLOADS[name][:3]
    key = ("ek60", C, P, S, i16)
    if key not in ctx.cache:
        for k in [k for k in ctx.cache if k != key]:
            del ctx.cache[k]
        ctx.free()
        d = (ctx.synth.ek60_device_i16 if i16 else ctx.synth.ek60_device)(C, P, S, seed=20260501, ss_every=ss_every)
        d["tau0"] = d["transmit_duration_nominal"][:, 0].contiguous()  # EK60 tau_eff = ping 0
        ctx.cache[key] = d
    d = ctx.cache[key]
    p = np.arange(P)
    ss = np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * (p // ss_every * ss_every) / 1e5), (C, 1))
    d["sound_speed_indicative"] = torch.from_numpy(ss).cuda()
    return d


def run_ek60(ctx, name, variant, cpu):
    args, torch, ops, sharding = ctx.args, ctx.torch, ctx.ops, ctx.sharding
    C, P, S = WORKLOADS[name][:3]
    chain = name == "cfg3"
    i16 = variant.startswith("int16") and not chain
    bins_only = variant.endswith("bins") and not chain  # no Sv array out: what the input's width is worth (2 / 4 B/sample)
    k1 = variant in ("sv", "sv32") and not chain  # BASELINE configs[1] to the letter: the compute_Sv kernel alone (K1)
    ss_every = 2000 if variant == "ss2000" else args.ss_every
    dt = ctx.dt
    d = ek60_volume(ctx, name, ss_every, i16)
    ns = d["ping_time_ns"]
    bin_ns = 20_000_000_000
    e0, _ = sharding.global_time_grid(ns.cpu().numpy(), bin_ns)
    n_t = P // 20
    r_max = float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2)  # analytic for this recipe
    n_r = len(np.arange(0, r_max + 1.0, 1.0)) - 1
    sv = torch.empty((C, P, S), dtype=dt, device="cuda") if not (chain or bins_only) else None
    mvbs = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")

    def one_pass(timer):
        coef = _coef(ctx, d)
        bs = ops.time_bin_offsets(ns, e0, bin_ns, n_t)
        if timer is not None:
            timer.start()
        if chain:  # noise blocks of 20 pings x 50 samples, SNR 3 dB
            a2 = coef[..., 4].contiguous()
            _, _, nz = ops.sv_noise_fused(d["backscatter_r"], coef, a2, 20, 50, dtype=dt)
            ops.sv_denoise_mvbs(d["backscatter_r"], coef, a2, nz, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt,
                                want_noise=args.chain_outputs == "all")
        elif k1:
            ops.sv_power(d["backscatter_r"], coef, dtype=dt, want_range=False, out=sv)
        elif i16:
            ops.sv_mvbs_fused_i16(d["raw_i16"], d["n_valid"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mvbs,
                                  want_sv=not bins_only)
        else:
            ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mvbs,
                              want_sv=not bins_only)
        if timer is not None:
            timer.stop()

    passes = ctx.passes(name)
    elapsed, kernel_ms = ctx.timed(one_pass, passes)
    n = C * P * S
    bps = BYTES_PER_SAMPLE[ctx.dtype] - (2 if i16 else 0) - ((8 if ctx.dtype == "float64" else 4) if bins_only else 0)
    if chain:  # two sweeps over the 4-B input + Sv, Sv_corrected (+ Sv_noise) out: SURVEY 8d line E = 32 / 20 B
        bps = 2 * BYTES_PER_SAMPLE[ctx.dtype] + ((BYTES_PER_SAMPLE[ctx.dtype] - 4) if args.chain_outputs == "all" else 0)
    key = f"{name}:{ctx.dtype}" + (":int16" if i16 else "") + (":bins" if bins_only else "") + (":sv" if k1 else "") + \
        (":corrected" if chain and args.chain_outputs != "all" else "") + (f":ss{ss_every}" if chain and ss_every != 1 else "")
    what = ("compute_Sv alone (K1, echo_range left lazy), Sv out" if k1 else
            "fused compute_Sv->compute_MVBS(20s x 1m), MVBS out only" if bins_only else
            "fused compute_Sv->compute_MVBS(20s x 1m), Sv+MVBS out" if not chain else
            "compute_Sv->remove_background_noise(20x50,3dB)->compute_MVBS(20s x 1m) in two sweeps, Sv+"
            + ("Sv_noise+" if args.chain_outputs == "all" else "") + "Sv_corrected+MVBS out")
    return line(ctx, samples_per_pass=n, passes=passes, elapsed=elapsed, scaling="weak",
                metric=("range-samples/sec through compute_Sv" if k1 else METRIC + (" with remove_background_noise" if chain else "")),
                workload=f"{name}: EK60 CW {C}x{P}x{S}, {what}" + (", int16 samples in" if i16 else ""),
                config={"sound_speed_changes_every_n_pings": ss_every, "sharding": "one GPU", "collective": "none"}, cpu=cpu,
                roofline=roofline("sv_noise_fast_kernel + sv_denoise_mvbs_{uniform,fast}_kernel" if chain
                                  else "sv_power_piece_kernel" if k1 else "fused_sv_mvbs_kernel", kernel_ms, n * bps, bps,
                                  traffic_key=key))


def run_api_pcie(ctx, cpu):
    """``api:pcie``: the two reference calls on a plain HOST echodata (NumPy arrays, nothing resident, no
    ``EchoData.to_device``) -- what a caller who hands over host buffers gets: every pass uploads the raw samples
    (4 B/sample over PCIe, pageable memory) before the kernel can run, and reads the MVBS back; Sv stays in HBM.  Never
    the headline's ``value`` (inputs resident is the contract); reported beside it as ``also_pcie``."""
    import logging

    import echopype_amd as ep

    C, P, S = 4, 100_000, 2000
    ctx.cache.clear()
    ctx.free()
    d = ctx.synth.ek60_numpy(C, 4, 8)
    h = ctx.synth.ek60_params(C, P, ss_every=ctx.args.ss_every)
    for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
              "absorption_indicative"):
        d[k] = h[k]
    d["ping_time"] = h["ping_time"]
    d["backscatter_r"] = ctx.synth.ek60_device(C, P, S, seed=20260511, ss_every=ctx.args.ss_every)["backscatter_r"].cpu().numpy()
    ed = ep.echodata.from_ek60_arrays(d)  # host arrays
    dtype = ctx.dtype

    def one_pass(timer):
        if timer is not None:
            timer.start()
        ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
        mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
        host = mv["Sv"].values  # the MVBS grid back on the host
        if timer is not None:
            timer.stop()
        return host.shape

    logging.disable(logging.WARNING)
    try:
        elapsed, region_ms = ctx.timed(one_pass, 1)
    finally:
        logging.disable(logging.NOTSET)
    n = C * P * S
    bps = BYTES_PER_SAMPLE[dtype]
    out = line(ctx, samples_per_pass=n, passes=1, elapsed=elapsed, scaling="weak", cpu=cpu,
               workload=f"api:pcie: EK60 CW {C}x{P}x{S} on HOST arrays (no to_device): compute_Sv(echodata) then "
                        "compute_MVBS(ds_Sv), MVBS read back; PCIe-inclusive, never the headline",
               config={"sharding": "one GPU", "collective": "none", "h2d_bytes_per_sample": 4,
                       "h2d_GBps": n * 4 / (elapsed / ctx.args.steps) / 1e9},
               roofline=roofline("fused_sv_mvbs_kernel behind the upload of its input", region_ms, n * bps, bps,
                                 traffic_key=None, note="region = upload + both calls + MVBS read-back; bound by PCIe, "
                                                        "not HBM: the fraction says how far from the resident rate"))
    return out


def run_api(ctx, cpu, variant=""):
    """``api:chain``: the reference's THREE calls of the chain on the cfg3 volume -- compute_Sv, remove_background_noise,
    compute_MVBS of the dataset with Sv := Sv_corrected -- file after file: two passes over the raw samples (32 B per
    sample in fp64), the second one inside compute_MVBS (clean.api.DenoiseSource).  Otherwise:
    The reference's own two calls on the cfg2 volume through the drop-in Dataset API, echodata resident in HBM
    (EchoData.to_device): one pass = calibrate.compute_Sv(echodata) then commongrid.compute_MVBS(ds_Sv, '1m', '20s').
    compute_Sv leaves Sv (and echo_range) to the first reader; compute_MVBS, reading first, writes the Sv array and the
    bins in ONE pass over the raw samples: 4 B in + 8 B out per sample for both calls (fp64).  Also timed: the same two
    calls with EPA_DEFER_SV=0 (K1, then the binning kernel on the Sv array: 20 B/sample) and the one-call compute_Sv_MVBS."""
    import logging

    import echopype_amd as ep

    if variant == "pcie":
        return run_api_pcie(ctx, cpu)
    C, P, S = WORKLOADS["api"][:3]
    raw = ek60_volume(ctx, "cfg2", ctx.args.ss_every)["backscatter_r"]
    d = ctx.synth.ek60_numpy(C, 4, 8)
    for k, v in list(d.items()):
        if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == (C, 4):
            d[k] = np.repeat(v[:, :1], P, axis=1)
    p = np.arange(P)
    d["sound_speed_indicative"] = np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1))
    d["backscatter_r"] = ep.DeviceArray(raw)
    d["ping_time"] = ctx.synth.T0 + (p * 1_000_000_000).astype("timedelta64[ns]")
    ed = ep.echodata.from_ek60_arrays(d).to_device()  # samples AND per-ping parameters resident in HBM
    dtype = ctx.dtype

    def two_calls():
        ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
        mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
        assert ds["Sv"].data.tensor is not None  # (the Sv array exists when the two calls return)
        return ds, mv

    def three_calls_of_the_chain():
        ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
        ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50)
        corrected = ds.copy()
        corrected["Sv"] = ds["Sv_corrected"]
        return ds, ep.commongrid.compute_MVBS(corrected, range_bin="1m", ping_time_bin="20s")

    chain = variant == "chain"
    held = []  # the previous pass's result: read (the deferred MVBS dataset assembled) after this pass's launch

    def one_pass(timer):
        if timer is not None:
            timer.start()
        r = three_calls_of_the_chain() if chain else two_calls()
        if timer is not None:
            timer.stop()
        held.append(r)
        while len(held) > 1:
            held.pop(0)[1]["Sv"].shape

    def finish():
        while held:
            held.pop(0)[1]["Sv"].shape

    logging.disable(logging.WARNING)
    try:
        passes = ctx.passes("api")
        elapsed, region_ms = ctx.timed(one_pass, passes, finish=finish)
        if chain:
            n = C * P * S
            bps = 32 if dtype == "float64" else 20  # raw in twice, Sv, Sv_noise, Sv_corrected out
            return line(ctx, samples_per_pass=n, passes=passes, elapsed=elapsed, scaling="weak", cpu=cpu,
                        workload=f"api:chain: EK60 CW {C}x{P}x{S}, calibrate.compute_Sv(echodata), "
                                 "clean.remove_background_noise(ds, 20, 50), commongrid.compute_MVBS(ds with Sv := "
                                 "Sv_corrected) through the Dataset API, echodata resident in HBM, pass 2 deferred to "
                                 "compute_MVBS, each result read after the next file's launch",
                        config={"sharding": "one GPU", "collective": "none"},
                        roofline=roofline("sv_noise_fast_kernel + sv_denoise_mvbs_* inside the three calls", region_ms,
                                          n * bps, bps, traffic_key=f"api:chain:{dtype}",
                                          note="region = the three API calls of one file incl. host work"))

        def med(f, prep=None):
            ts = []
            for _ in range(4):
                a = prep() if prep else None
                ctx.torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = f(a) if prep else f()
                ctx.torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                del r, a
            return float(np.median(ts[1:])) * 1e3

        sv_ms = med(lambda: ep.calibrate.compute_Sv(ed, dtype=dtype))
        mv_ms = med(lambda ds: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s"),
                    prep=lambda: ep.calibrate.compute_Sv(ed, dtype=dtype))
        one_call = med(lambda: ep.compute_Sv_MVBS(ed, range_bin="1m", ping_time_bin="20s", dtype=dtype))

        def three_calls():  # the chain as the reference's user writes it; the MVBS is that of Sv_corrected
            ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
            ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50)
            corrected = ds.copy()
            corrected["Sv"] = ds["Sv_corrected"]
            return ds, ep.commongrid.compute_MVBS(corrected, range_bin="1m", ping_time_bin="20s")

        chain_ms = med(three_calls)

        def chain_pipelined(n_files=6):  # file after file, each MVBS read after the next file's launch
            ctx.torch.cuda.synchronize()
            t0 = time.perf_counter()
            prev = None
            for _ in range(n_files):
                cur = three_calls()
                if prev is not None:
                    prev[1]["Sv"].shape
                prev = cur
            prev[1]["Sv"].shape
            ctx.torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n_files * 1e3

        chain_pipelined()
        chain_pipe_ms = chain_pipelined()
        os.environ["EPA_DEFER_SV"] = "0"
        try:
            eager_ms = med(two_calls)
            chain_eager_ms = med(three_calls)
        finally:
            del os.environ["EPA_DEFER_SV"]
    finally:
        logging.disable(logging.NOTSET)
    n = C * P * S
    bps = BYTES_PER_SAMPLE[dtype]  # raw in, Sv out: compute_MVBS writes the deferred Sv in its own pass
    return line(ctx, samples_per_pass=n, passes=passes, elapsed=elapsed, scaling="weak", cpu=cpu,
                workload=f"api: EK60 CW {C}x{P}x{S}, calibrate.compute_Sv(echodata) then commongrid.compute_MVBS(ds_Sv) "
                         "through the Dataset API, echodata resident in HBM, Sv deferred to compute_MVBS's pass, each "
                         "result read after the next pass's launch",
                config={"compute_Sv_ms": sv_ms, "compute_MVBS_ms": mv_ms, "one_call_compute_Sv_MVBS_ms": one_call,
                        "two_calls_not_deferred_ms": eager_ms, "chain_three_calls_ms": chain_ms,
                        "chain_three_calls_file_after_file_ms": chain_pipe_ms,
                        "chain_three_calls_not_deferred_ms": chain_eager_ms, "sharding": "one GPU", "collective": "none"},
                roofline=roofline("fused_sv_mvbs_kernel inside compute_MVBS (+ host parameter selection)", region_ms, n * bps,
                                  bps, traffic_key=f"api:{dtype}", note="region = both API calls incl. host work"))


# ---------------------------------------------------------------------------------------- next:*
def run_next(ctx, variant, cpu):
    """The "next" rows of SURVEY 8f on the driver's record, each through its API entry point on a resident EK60 dataset
    (4 x 100 000 x 2000, a new sound speed at every ping -- so every ping has its own range / depth vector):
      depth  compute_Sv -> consolidate.add_depth(ds, depth_offset) -> commongrid.compute_MVBS(ds, range_var="depth")
             (consolidate/api.py:68-243, commongrid/api.py:30-191): one sweep, raw in + Sv out = 12 B/sample (depth stays
             lazy); ``depthw``: depth written by the same sweep, 20 B/sample
      masks  clean.mask_impulse_noise / mask_attenuated_signal / mask_transient_noise on ``depth`` + mask.apply_mask of the
             three (clean/api.py:30-359, mask/api.py:307-464): 3 x (8 + 8 + 1) + (8 + 3 + 8) = 70 B/sample
      nasc   commongrid.compute_NASC(ds) (commongrid/api.py:269-416): Sv + depth read, 16 B/sample
    (int16 ingest, the fourth row, is the ``cfg2:int16`` line.)  ``roofline`` = the HIP-event bracket round the calls of
    one pass against those algorithmic bytes."""
    import logging

    import echopype_amd as ep

    C, P, S = WORKLOADS["next"][:3]
    # masks2000: the recorded sound speed -- hence the range / depth vector -- changes every 2000 pings (an operator's
    # setting holds for a while) instead of at every ping, the worst case the other rows use
    # masksidx: the masks that have an index-binned form (SURVEY 8f rank 2 names those) with use_index_binning=True
    ss_every = 2000 if variant == "masks2000" else 1
    by_index = variant == "masksidx"
    tag = variant
    if variant in ("masks2000", "masksidx"):
        variant = "masks"
    key = ("next", C, P, S, ss_every)
    if key not in ctx.cache:
        for k in [k for k in ctx.cache if k != key]:
            del ctx.cache[k]
        ctx.free()
        d = ctx.synth.ek60_numpy(C, 4, 8)
        h = ctx.synth.ek60_params(C, P, ss_every=ss_every)
        for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
                  "absorption_indicative"):
            d[k] = h[k]
        d["ping_time"] = h["ping_time"]
        d["backscatter_r"] = ep.DeviceArray(ctx.synth.ek60_device(C, P, S, seed=20260509, ss_every=ss_every)["backscatter_r"])
        ctx.cache[key] = ep.echodata.from_ek60_arrays(d).to_device()
    ed = ctx.cache[key]
    dtype = ctx.dtype
    logging.disable(logging.WARNING)
    try:
        finish = None
        if variant in ("depth", "depthw"):
            # depth: lazy (an affine function of the coefficient rows, binned inside the pass that writes Sv);
            # depthw: the depth array written by the same pass as well (EPA_DEPTH_WITH_MVBS=1)
            held = []  # the previous pass's result: read (the deferred MVBS dataset assembled) after this pass's launch
            written = variant == "depthw"

            def one_pass(timer):
                if timer is not None:
                    timer.start()
                ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
                ds = ep.consolidate.add_depth(ds, depth_offset=5.0)
                mv = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="1m", ping_time_bin="20s")
                assert ds["Sv"].data.materialized and ds["depth"].data.materialized == written
                if timer is not None:
                    timer.stop()
                held.append(mv)
                while len(held) > 1:
                    held.pop(0)["Sv"].shape

            def finish():
                while held:
                    held.pop(0)["Sv"].shape

            if written:
                os.environ["EPA_DEPTH_WITH_MVBS"] = "1"
            f64 = dtype == "float64"
            bps = (20 if f64 else 12) if written else (12 if f64 else 8)
            kern = "fused_sv_mvbs_kernel<.., DEPTH> inside compute_MVBS: the three calls are one sweep of the raw samples"
            what = ("compute_Sv -> add_depth(depth_offset=5) -> compute_MVBS(range_var='depth', 1m x 20s)"
                    + (", depth written by the same pass" if written else ", depth left lazy")
                    + ", each result read after the next pass's launch")
        else:
            ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
            ds = ep.consolidate.add_depth(ds, depth_offset=5.0)
            ds["Sv"].data.tensor, ds["depth"].data.tensor  # (resident arrays: the rows below start from an Sv dataset)
            if variant == "masks":
                def one_pass(timer):
                    if timer is not None:
                        timer.start()
                    m1 = ep.clean.mask_impulse_noise(ds, depth_bin="5m", num_side_pings=2, impulse_noise_threshold="10.0dB",
                                                     use_index_binning=by_index)
                    m2 = ep.clean.mask_attenuated_signal(ds, upper_limit_sl="150.0m", lower_limit_sl="250.0m",
                                                         num_side_pings=15, attenuation_signal_threshold="8.0dB")
                    m3 = ep.clean.mask_transient_noise(ds, func="nanmean", depth_bin="10m", num_side_pings=25,
                                                       exclude_above="20.0m", transient_noise_threshold="12.0dB",
                                                       use_index_binning=by_index)
                    out = ep.mask.apply_mask(ds, [m1, m2, m3])
                    if timer is not None:
                        timer.stop()
                    return out
                bps = 70 if dtype == "float64" else 38
                kern = "range_bin_smooth + impulse_compare + attenuated_* + pool_value_* + mask_and + apply_mask kernels"
                what = ("mask_impulse_noise + mask_attenuated_signal + mask_transient_noise (on depth) + apply_mask"
                        + (", use_index_binning=True" if by_index else ""))
            elif variant == "nasc":
                p = np.arange(P)
                ds["latitude"] = (("ping_time",), 45.0 + 1e-5 * p)
                ds["longitude"] = (("ping_time",), -125.0 + 2e-5 * p)

                def one_pass(timer):
                    if timer is not None:
                        timer.start()
                    out = ep.commongrid.compute_NASC(ds, range_bin="10m", dist_bin="0.5nmi")
                    if timer is not None:
                        timer.stop()
                    return out
                bps, kern = 16 if dtype == "float64" else 8, "nasc_accumulate + nasc_finalize kernels"
                what = "compute_NASC(range_bin='10m', dist_bin='0.5nmi')"
            else:
                sys.exit(f"unknown next:{variant}")
        passes = ctx.passes("next")
        elapsed, region_ms = ctx.timed(one_pass, passes, finish=finish)
    finally:
        os.environ.pop("EPA_DEPTH_WITH_MVBS", None)
        logging.disable(logging.NOTSET)
    n = C * P * S
    return line(ctx, samples_per_pass=n, passes=passes, elapsed=elapsed, scaling="weak", cpu=cpu,
                metric="range-samples/sec through the SURVEY 8f row",
                workload=f"next:{tag}: EK60 CW {C}x{P}x{S} Sv dataset resident in HBM, "
                         f"{what} through the Dataset API",
                config={"sharding": "one GPU", "collective": "none", "sound_speed_changes_every_n_pings": ss_every},
                roofline=roofline(kern, region_ms, n * bps, bps,
                                  traffic_key=f"next:{tag}:{dtype}",
                                  note="region = the API calls of one pass incl. host work"))


# ---------------------------------------------------------------------------------------- EK80 BB: cfg4
def run_ek80(ctx, name, variant, cpu):
    """EK80 BB complex -> pulse compression + Sv (epa_sv_complex_fft) -> MVBS.  The per-(channel, ping) parameter rows
    and the replicas are assembled once by the drop-in's own calibrator (host, O(C*P)); one pass = the two kernels over
    the resident planes.  ``planes64``: backscatter_r / _i float64 as the converter stores them (parse_base.py:306-309)."""
    import echopype_amd as ep
    from echopype_amd import _lib
    from echopype_amd.calibrate.api import CALIBRATOR

    torch, ops, sharding, synth = ctx.torch, ctx.ops, ctx.sharding, ctx.synth
    C, P, S = WORKLOADS[name][:3]
    B = 4
    pdt = torch.float64 if variant == "planes64" else torch.float32
    key = ("ek80", C, P, S, pdt)
    if key not in ctx.cache:
        ctx.cache.clear()
        ctx.free()
        g = torch.Generator(device="cuda")
        g.manual_seed(20260504)
        re = torch.empty((C, P, S, B), dtype=pdt, device="cuda")
        im = torch.empty((C, P, S, B), dtype=pdt, device="cuda")
        slab = max(1, P // 50)
        for p0 in range(0, P, slab):  # in slabs: no second copy of the planes
            n = min(slab, P - p0)
            re[:, p0:p0 + n] = torch.randn((C, n, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
            im[:, p0:p0 + n] = torch.randn((C, n, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
        nan_pings = torch.rand(P, generator=g, device="cuda") < 0.10
        tail = int(round(0.05 * S))
        re[:, nan_pings, S - tail:] = float("nan")
        im[:, nan_pings, S - tail:] = float("nan")
        ctx.cache[key] = (re, im)
    re, im = ctx.cache[key]
    d = synth.ek80_numpy(C, 4, 64, B)  # parameters only; the sample planes are generated on the device
    pidx = np.arange(P)
    ping_time = synth.T0 + (pidx * 1_000_000_000).astype("timedelta64[ns]")
    d.update(backscatter_r=ep.DeviceArray(re), backscatter_i=ep.DeviceArray(im), sample_interval=np.full((C, P), 8e-6),
             sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * pidx / 1e5), (C, 1)), ping_time=ping_time)
    ed = ep.echodata.from_ek80_arrays(d, synth.ek80_filters())
    cal = CALIBRATOR["EK80"](ed, env_params=None, cal_params=None, ecs_file=None, waveform_mode="BB",
                             encode_mode="complex", dtype=ctx.dtype, device=None)
    k, _ = cal._complex_inputs("Sv")
    ns = torch.from_numpy(ping_time.astype(np.int64)).cuda()
    bin_ns = 20_000_000_000
    e0, _ = sharding.global_time_grid(ns.cpu().numpy(), bin_ns)
    n_t = P // 20
    range_bin = 0.1
    rmax = float((S - 1) * 8e-6 * 1500.5 / 2)
    n_r = len(np.arange(0, rmax + range_bin, range_bin)) - 1
    rows = ops.power_rows_of_complex(k["ccoef"])

    def one_pass(timer):
        bs = ops.time_bin_offsets(ns, e0, bin_ns, n_t)
        if timer is not None:
            timer.start()
        res = ops.sv_complex(k["re"], k["im"], k["ccoef"], replica=k["replica"], replica_off=k["replica_off"],
                             max_taps=k["max_taps"], dtype=ctx.dt, want_range=False)
        if timer is not None:
            timer.stop()
        ops.mvbs(res["out"], bs, n_t, range_bin, n_r, coef=rows)

    passes = ctx.passes(name)
    elapsed, kernel_ms = ctx.timed(one_pass, passes)
    n = C * P * S
    esz = 8 if variant == "planes64" else 4
    bps = B * 2 * esz + (8 if ctx.dtype == "float64" else 4)  # SURVEY 8d line F: r/i sectors in, Sv out
    taps = int(k["max_taps"])
    return line(ctx, samples_per_pass=n, passes=passes, elapsed=elapsed, scaling="weak", cpu=cpu,
                workload=f"{name}: EK80 BB {C}x{P}x{S}x{B} sectors, {'float64' if esz == 8 else 'float32'} r/i planes, "
                         f"{taps}-tap replica: pulse compression + Sv, then MVBS(20s x {range_bin}m)",
                config={"sample": "one (channel, ping, range_sample) output", "sharding": "one GPU", "collective": "none",
                        "fft_dtype": "complex128" if ctx.dtype == "float64" else "complex64"},
                roofline=roofline("sv_complex_fft_kernel", kernel_ms, n * bps, bps,
                                  traffic_key=f"{name}:{ctx.dtype}" + (":planes64" if esz == 8 else ""),
                                  direct_form_tflops=8.0 * taps * n / (kernel_ms * 1e-3) / 1e12))


# ---------------------------------------------------------------------------------------- cfg5: tiles, N >= 1
class Cfg5:
    """BASELINE configs[4]: C x P_total x S as resident tiles of <= tile_pings pings, dealt to the ranks round-robin
    (tile g lives on rank g % N): tiles j N .. j N + N - 1 are a contiguous dataset sharded by ping_time over the N
    ranks.  ``layout(offset_ns)``: the ops-level harness -- time grid, exchange plan and one pass of the kernels for ping
    times shifted by ``offset_ns`` against the 20-s grid (0: tile edges on bin edges, no exchange; 10 s: every tile edge
    cuts a bin).  ``api_layout(offset_ns)``: the same tiles as EchoData objects through the product entry points."""

    BIN_NS = 20_000_000_000

    def __init__(self, ctx, C, P_total, S, tile_pings=None, ss_every=1, range_bin=1.0):
        torch, sharding, synth = ctx.torch, ctx.sharding, ctx.synth
        self.ctx, self.C, self.S, self.P_total, self.range_bin, self.ss_every = ctx, C, S, P_total, range_bin, ss_every
        if tile_pings is None:  # the tiling belongs to the WORKLOAD (eight tiles of the whole volume), not to the split
            n_tiles = -(-8 // ctx.world) * ctx.world  # (a multiple of the rank count: every rank joins every dataset)
            tile_pings = min(TILE_PINGS, -(-P_total // (20 * n_tiles)) * 20)
        self.tile_p = tile_pings
        self.n_tiles = -(-P_total // tile_pings)
        self.gtiles = list(range(ctx.rank, self.n_tiles, ctx.world))
        self.spans = [(g * tile_pings, min(P_total, (g + 1) * tile_pings)) for g in self.gtiles]
        self.tiles = []
        for a, b in self.spans:
            d = synth.ek60_device(C, b - a, S, seed=20260505 + a // 20, ping0=a, ss_every=ss_every)
            d["tau0"] = torch.full((C,), 1.024e-3, dtype=torch.float64, device="cuda")  # ping 0 of the WHOLE file
            self.tiles.append(d)
        self.n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + range_bin, range_bin)) - 1
        dt = ctx.dt
        esz = 8 if ctx.dtype == "float64" else 4
        free_b = torch.cuda.mem_get_info()[0]
        # Sv of every tile resident?  (not when the ranks share one device: each sees the same free memory)
        self.keep_all = sum((b - a) for a, b in self.spans) * C * S * esz < free_b - (8 << 30) and \
            not getattr(ctx.args, "single_device", False)
        self.sv = None
        self.shard = sharding.ShardContext()

    def sv_buffers(self):
        """Ops-level harness: preallocated Sv outputs (one per tile when they all fit, else one reused buffer)."""
        torch, C, S, dt = self.ctx.torch, self.C, self.S, self.ctx.dt
        if self.sv is None:
            if self.keep_all:
                self.sv = [torch.empty((C, b - a, S), dtype=dt, device="cuda") for a, b in self.spans]
            else:  # one buffer, every tile writes its Sv through a contiguous view of it
                buf = torch.empty((C, self.tile_p, S), dtype=dt, device="cuda")
                self.sv = [buf.view(-1)[:C * (b - a) * S].view(C, b - a, S) for a, b in self.spans]
        return self.sv

    def echodata(self, i, offset_ns):
        """Tile ``i`` as the EchoData the converter would hand over (convert/set_groups_ek60.py), samples AND per-ping
        parameters resident in HBM (EchoData.to_device: the analogue of persisting the reference's dask arrays)."""
        import echopype_amd as ep

        synth, C = self.ctx.synth, self.C
        a, b = self.spans[i]
        d = synth.ek60_numpy(C, 4, 8)  # channel names, per-channel values, the pulse-length tables
        h = synth.ek60_params(C, b - a, ping0=a, ss_every=self.ss_every)
        for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
                  "absorption_indicative"):
            d[k] = h[k]
        d["ping_time"] = h["ping_time"] + np.timedelta64(int(offset_ns), "ns")
        d["backscatter_r"] = ep.DeviceArray(self.tiles[i]["backscatter_r"])
        return ep.echodata.from_ek60_arrays(d, source_file=f"synthetic_ek60_tile{self.gtiles[i]:02d}.raw").to_device()

    def api_layout(self, offset_ns, lag=1):
        """(one_pass, finish, state): one pass = every resident tile through the product entry points -- world 1: the
        reference's two calls; world > 1: sharding.compute_Sv_MVBS on the tile as this rank's shard of dataset j.  The
        result of a tile is READ (the deferred MVBS dataset assembled: three doubles come back from the GPU) after the
        next ``lag`` tiles have been launched, then dropped (its Sv array goes back to the allocator)."""
        import logging

        import echopype_amd as ep

        ctx, sharding = self.ctx, self.ctx.sharding
        eds = [self.echodata(i, offset_ns) for i in range(len(self.tiles))]
        dtype = ctx.dtype
        rb = f"{self.range_bin:g}m"
        if ctx.world == 1 and not getattr(ctx.args, "sharded_at_1", False):
            def call(ed):
                ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
                return ds, ep.commongrid.compute_MVBS(ds, range_bin=rb, ping_time_bin="20s")
        else:
            shard = sharding.MVBSShard()
            tau0 = np.full(self.C, 1.024e-3)  # EK60 tau_effective = ping 0 of the WHOLE file (calibrate_ek.py:154-162)

            def call(ed):
                return sharding.compute_Sv_MVBS(ed, range_bin=rb, ping_time_bin="20s", dtype=dtype, shard=shard,
                                                tau_effective_first_ping=tau0)
        torch = ctx.torch
        n_streams = max(1, int(getattr(ctx, "tile_streams", 1) or 1))
        # The loop itself is the product's: ``echopype_amd.pipeline`` deals consecutive tiles to ``n_streams`` side
        # streams and hands a tile's result out -- the deferred MVBS dataset assembled: three doubles come back from the
        # GPU -- after the next ``lag`` tiles have been launched.  ONE pipeline runs through all the passes of a timed
        # region (no drain at a pass boundary); ``finish`` reads what is still in flight.
        # host_s / calls: wall time the HOST spends inside the entry-point calls of a tile (launches, control-plane
        # messages; no wait for the GPU on the deferred routes) -- what has to stay under the kernel's ~9 ms per tile for
        # the GPU to run back to back, and the per-call cost that decides the scaling at N = 8 (one tile per rank and pass)
        state = {"last": None, "n_read": 0, "host_s": 0.0, "calls": 0, "timing": False, "timers": [], "pipe": None,
                 "launched": 0}
        T = len(eds)

        def launch(i):
            # the HIP-event bracket goes round the tiles from pass to pass: roofline.kernel_ms is the mean over ALL tiles
            # of the volume (the calls of a tile behind a busy stream), not the first tile's
            timed = state["timing"] and (i % T) == (i // T) % T
            tm = None
            if timed:
                tm = ctx.ops.Timer()
                tm.start()
            t_host = time.perf_counter()
            item = call(eds[i % T])
            if state["timing"]:
                state["host_s"] += time.perf_counter() - t_host
                state["calls"] += 1
            if tm is not None:
                tm.stop()
                state["timers"].append(tm)
            return item

        def read(item):
            ds, mv = item
            state["last"] = (tuple(mv["Sv"].shape),)
            state["n_read"] += 1

        def one_pass(timer):
            logging.disable(logging.WARNING)  # (the NaN-coordinate warning of every tile: 10 % of the pings are padded)
            try:
                if state["pipe"] is None:
                    state["timing"] = timer is not None
                    state["pipe"] = ep.pipeline.Pipeline(launch, streams=n_streams if n_streams > 1 else 0, lag=lag)
                for _ in range(T):
                    for item in state["pipe"].submit(state["launched"]):
                        read(item)
                    if not os.environ.get("EPA_BENCH_KEEP"):  # (development knob)
                        item = None  # (a result that has been read is not kept over the next launch: its Sv array is 32.8 GB)
                    state["launched"] += 1
            finally:
                logging.disable(logging.NOTSET)

        def finish():
            logging.disable(logging.WARNING)
            try:
                if state["pipe"] is not None:
                    for item in state["pipe"].drain():
                        read(item)
                state["pipe"] = None
            finally:
                logging.disable(logging.NOTSET)

        return one_pass, finish, state, eds

    def layout(self, offset_ns):
        ctx, torch, ops, sharding = self.ctx, self.ctx.torch, self.ctx.ops, self.ctx.sharding
        C, n_r, dt = self.C, self.n_r, ctx.dt
        ts = [d["ping_time_ns"] + offset_ns for d in self.tiles]
        ends = np.array([int(x) for t in ts for x in (t[0], t[-1])], dtype=np.int64)
        e0, _ = sharding.global_time_grid(ends, self.BIN_NS)  # ONE grid for every tile of every rank
        info = []
        for t in ts:
            f, l = sharding.local_bin_span(t[[0, -1]].cpu().numpy(), e0, self.BIN_NS)
            info.append((t, e0 + f * self.BIN_NS, f, l, l - f + 1))
        plan = self.shard.plan([(f, l) for _, _, f, l, _ in info], C, n_r, "cuda")
        mv = [torch.empty((C, n, n_r), dtype=dt, device="cuda") for *_, n in info]
        dst = {(i, w): mv[i][:, 0 if w == 0 else -1] for i in range(len(mv)) for w in (0, 1)}
        tiles, sv, rb = self.tiles, self.sv_buffers(), self.range_bin

        turn = [0]

        def one_pass(timer):
            rows = {}
            timed_tile = turn[0] % len(tiles)  # (the bracket goes round the tiles: kernel_ms = mean over all of them)
            turn[0] += 1
            for i, d in enumerate(tiles):
                t, e0l, f, l, n_t = info[i]
                coef = _coef(ctx, d)
                bs = ops.time_bin_offsets(t, e0l, self.BIN_NS, n_t)
                if timer is not None and i == timed_tile:
                    timer.start()
                res = ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, rb, n_r, dtype=dt, sv_out=sv[i],
                                        mvbs_out=mv[i], want_partials=plan.shared)
                if timer is not None and i == timed_tile:
                    timer.stop()
                if plan.shared:
                    for w, r in sharding.mvbs_edge_rows(res["sum"], res["cnt"]).items():
                        rows[(i, w)] = r
            if plan.shared:  # ONE all-reduce for every cut bin of every tile of every rank; owners finalise their rows
                plan.merge_mvbs(rows, dst)

        return info, plan, mv, one_pass


def ranks_info(ctx):
    """world size, backend and the device every rank runs on, gathered over the process group: a SCALE record then
    proves RCCL saw N ranks on N devices."""
    torch, dist = ctx.torch, ctx.dist
    mine = f"{ctx.rank}:cuda{torch.cuda.current_device()}"
    name = torch.cuda.get_device_name()
    if ctx.world == 1:  # (--sharded-at-1: a one-rank group of the real backend)
        return {"world_size": 1, "backend": dist.get_backend() if dist.is_initialized() else "none", "devices": [mine],
                "device_name": name}
    got = [None] * ctx.world
    dist.all_gather_object(got, (mine, name), group=ctx.sharding.control_group())
    names = sorted({n for _, n in got})
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices": [m for m, _ in got],
            "device_name": names[0] if len(names) == 1 else names}


def run_cfg5(ctx, cpu, variant=""):
    """The headline: the eight tiles through the product entry points (api_layout); beside it the ops-level harness on
    the same resident tiles, with and without cut bins.  N = 1: the tiles -- independent datasets -- are dealt to
    ``--tile-streams`` HIP streams by the package's own loop, ``echopype_amd.pipeline.run`` (default 2: the kernel of a tile
    runs beside the next tile's; ``cfg5:one`` = the caller's stream, a launch at a time); N > 1 and --sharded-at-1 alike:
    a rank's consecutive shards alternate between the streams, the collectives are issued in item order.  N = 1: Sv of the tiles goes to one reused buffer (ops level) /
    to the allocator's recycled block (API): 131 GB in + 262 GB out does not fit 288 GB otherwise."""
    args, world = ctx.args, ctx.world
    # (--single-device: the ranks of the dry run share ONE GPU's 288 GB -- a second Sv in flight per rank does not fit)
    ctx.tile_streams = 1 if variant == "one" or args.single_device else max(1, args.tile_streams)
    C, _, S = WORKLOADS["cfg5"][:3]
    P_total = args.pings_total or WORKLOADS["cfg5"][1]
    job = Cfg5(ctx, C, P_total, S, tile_pings=args.tile_pings, ss_every=args.ss_every)
    passes = ctx.passes("cfg5")
    steps = args.steps
    # (a) ops level, bin-aligned layout: no bin is shared, no data collective
    _, plan_a, mv_a, pass_a = job.layout(0)
    assert not plan_a.shared
    el_a, km_a = ctx.timed(pass_a, passes)
    del mv_a, pass_a
    # (b) ops level, ping times 10 s off the grid: every tile edge cuts a bin -> edge exchange on the timed path
    _, plan_b, mv_b, pass_b = job.layout(10_000_000_000)
    single = world == 1 and len(job.spans) == 1
    assert plan_b.shared or single
    el_b, km_b = (ctx.timed(pass_b, passes) if not single else (el_a, km_a))
    edges_b, bytes_b = len(plan_b.edges), plan_b.nbytes
    del mv_b, pass_b, plan_b
    job.sv = None
    ctx.free()
    # (c) the product entry points on the same tiles, same 10-s offset: the headline
    lag = max(1, args.read_lag)
    pass_c, finish_c, state, eds = job.api_layout(10_000_000_000, lag=lag)
    elapsed, region_ms = ctx.timed(pass_c, passes, finish=finish_c, timers_of=lambda: state["timers"])
    assert state["n_read"] == len(job.tiles) * passes * (steps + args.warmup)  # every result was read
    if world == 1 and not getattr(args, "sharded_at_1", False):
        route = "compute_Sv -> compute_MVBS('1m','20s') per tile"
        coll = "none at 1 rank"
    else:
        route = "sharding.compute_Sv_MVBS(echodata_shard, shard=MVBSShard()) per tile"
        coll = (f"per dataset of {world} tiles: cut bins all_reduce(SUM) + range max all_reduce(MAX) in HBM over "
                f"{args.backend}; control scalars over gloo")
    # (ops_level_*: the kernels called directly on preallocated buffers, all cut bins of a rank's tiles exchanged)
    host_ms = state["host_s"] / max(1, state["calls"]) * 1e3
    if world > 1:  # the slowest rank's figure
        t = ctx.torch.tensor([host_ms], dtype=ctx.torch.float64)
        ctx.dist.all_reduce(t, op=ctx.dist.ReduceOp.MAX, group=ctx.sharding.control_group())
        host_ms = float(t.item())
    cfg = {"tiles": f"{job.n_tiles} x {job.tile_p} pings over {world} rank(s)",
           "route": route, "collective": coll, "results_read": f"{lag} tile(s) late", "host_ms_per_call": host_ms,
           "mvbs_shape_last_tile": list(state["last"][0]),
           "ops_level_ms_per_pass": el_b / steps / passes * 1e3, "ops_level_kernel_ms": km_b,
           "ops_level_edge_bins": edges_b, "allreduce_bytes": bytes_b,
           "ranks": ranks_info(ctx)}
    n_first = C * (job.spans[0][1] - job.spans[0][0]) * S if job.spans else 0
    bps = BYTES_PER_SAMPLE[ctx.dtype]
    del eds, pass_c, finish_c
    if ctx.rank != 0:
        return None
    k = ctx.tile_streams
    if k > 1:
        # Launches run side by side: the HIP-event bracket of ONE tile's calls spans the time it shares the GPU with its
        # neighbour.  The GPU is busy with this rank's launches and nothing else, so the duration a launch costs is the
        # timed region's wall time per launch (host gaps included: an upper bound); the bracket goes beside it.
        cfg["tile_streams"] = k
        extra = {"launch": f"wall per {job.tile_p}-ping tile, {k} side by side",
                 "kernel_ms_each": region_ms}
        region_ms = elapsed / steps / passes / max(1, len(job.tiles)) * 1e3
    else:
        extra = {"launch": f"API calls of one {job.tile_p}-ping tile (mean), HIP events"}
    return line(ctx, samples_per_pass=C * P_total * S, passes=passes, elapsed=elapsed, scaling="strong", cpu=cpu,
                workload=f"cfg5: EK60 CW {C}x{P_total}x{S} in ping_time tiles, compute_Sv -> compute_MVBS(20s x 1m), Sv+MVBS out",
                config=cfg,
                roofline=roofline("fused_sv_mvbs_kernel (+ K0 of the calls)", region_ms,
                                  n_first * bps, bps, traffic_key=f"cfg5api:{ctx.dtype}", **extra))


# ---------------------------------------------------------------------------------------- main
def relaunch(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU and pass its JSON line through."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def compact(x):
    """Floats to seven significant digits (the lines have to fit the driver's 2000-character tail)."""
    if isinstance(x, float):
        return float(f"{x:.7g}")
    if isinstance(x, dict):
        return {k: compact(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [compact(v) for v in x]
    return x


# what a line can lose, in this order, when it would not fit the driver's 2000-character tail (the last line carries the
# other lines' figures and grows with them); everything the contract names stays
DROP_ORDER = (("config", "ops_level_edge_bins"), ("config", "allreduce_bytes"), ("config", "ops_level_kernel_ms"),
              ("config", "ops_level_ms_per_pass"), ("config", "mvbs_shape_last_tile"), ("config", "results_read"),
              ("roofline", "traffic_source"), ("cpu_baseline", "multicore_sample"), ("config", "route"),
              ("roofline", "launch"), ("config", "also_cfg4"), ("config", "also_cfg3"), ("config", "collective"))


def fit(out, limit=1950):
    """The line as compact JSON, optional keys dropped (the first ``config.dropped`` of DROP_ORDER) until it fits."""
    out = compact(out)
    dropped = []
    for sec, key in DROP_ORDER:
        txt = json.dumps(out, separators=(",", ":"))
        if len(txt) <= limit:
            return txt
        if key in out.get(sec, {}):
            del out[sec][key]
            dropped.append(key)
            out["config"]["dropped"] = len(dropped)
    return json.dumps(out, separators=(",", ":"))


def summary(out):
    """'<Gsamp/s> f<fraction of 8 TB/s>' of a line, for the headline's also_* strings."""
    g = out["value"] / 1e9
    return (f"{g:.0f}" if g >= 100 else f"{g:.1f}") + f" f{out['roofline']['frac']:.3f}"


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch(args))
    if args.single_device:
        local_rank = 0
    if world > 1:
        todo = [args.workload or "cfg5"]
        if todo != ["cfg5"]:
            sys.exit("N > 1 runs the ping-sharded cfg5 workload")
    elif args.workload:
        todo = args.workload.split(",")  # (several lines in one process, in this order: "cfg5:one,cfg5")
    else:
        todo = ["cfg5"] if args.only_headline else list(DEFAULT_LINES)

    def kind_of(w):
        if w == "api:chain":
            return "chain"
        return {"cfg3": "chain", "cfg4": "bb", "cfg4small": "bb"}.get(w.partition(":")[0], "ek60")

    # CPU baselines first (rank 0): they fork worker processes, which must happen before HIP is initialised
    cpu = {}
    if rank == 0 and not args.no_cpu_baseline:
        for w in todo:
            kind = kind_of(w)
            if kind not in cpu:
                cpu[kind] = (cpu_baseline_bb() if kind == "bb" else
                             cpu_baseline_ek60(chain=kind == "chain", multicore=kind == "ek60"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL across processes needs dmabuf IPC on this host driver)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (echopype_amd sets it too: the pipeline's two streams need queues of their own)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    if world == 1 and args.sharded_at_1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=0, world_size=1)
    ctx = Ctx(args, world, rank)
    also = {}
    for i, spec in enumerate(todo):
        w, _, variant = spec.partition(":")
        if w not in WORKLOADS:
            sys.exit(f"unknown workload {w!r}")
        ctx.dtype = DT.get(variant, args.dtype)
        if w == "next" and not variant:
            sys.exit("next:depth | next:depthw | next:masks | next:masks2000 | next:masksidx | next:nasc")
        if w == "cfg5":
            ctx.cache.clear()
            ctx.free()
            out = run_cfg5(ctx, cpu.get("ek60"), variant)
        elif w == "api":
            out = run_api(ctx, cpu.get("chain" if variant == "chain" else "ek60"), variant)
        elif w.startswith("cfg4"):
            out = run_ek80(ctx, w, variant, cpu.get("bb"))
        elif w == "next":
            out = run_next(ctx, variant, cpu.get("ek60"))
        else:
            out = run_ek60(ctx, w, variant, cpu.get(kind_of(w)))
        if rank == 0 and out is not None:
            if i == len(todo) - 1 and also:  # the parsed line carries the others' key figures (flat, short strings)
                out["config"].update(also)
                out["config"]["also_unit"] = "Gsamp/s f<frac>"
            fam, _, var = spec.partition(":")  # one string per workload family: "441.4 f0.668; f32 628.0 f0.636"
            key = "also_" + fam
            if spec == "api:pcie":  # (its own key: the PCIe-inclusive rate must not be mistaken for a resident line)
                key, var = "also_pcie", ""
            if spec not in ("cfg2:int16f32", "cfg2:bins", "cfg2:int16bins"):  # (own lines only: the headline has 2000 characters)
                also[key] = (also[key] + "; " if key in also else "") + (var + " " if var else "") + summary(out)
            txt = fit(out)  # (no padding: the headline carries the other lines' figures)
            print(txt, flush=True)
            if args.out:
                with open(args.out, "a") as f:
                    f.write(txt + "\n")
        ctx.free()
    if world > 1 or dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
