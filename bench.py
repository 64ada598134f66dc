#!/usr/bin/env python3
"""bench.py -- range-samples/s through compute_Sv -> compute_MVBS on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ...]

N = 1 (default).  The HEADLINE line (printed LAST) is BASELINE configs[1]: synthetic EK60 CW, 4 channels x 500 000
pings x 2000 range samples (4.0 G samples) through the metric's path compute_Sv -> compute_MVBS (20 s x 1 m bins), fp64,
generated in HBM.  One STEP = the whole hot path over the resident volume:
    epa_power_coef_ek (K0) -> epa_time_bin_offsets -> epa_sv_mvbs_fused (K1+K5: writes Sv f64 and the MVBS grid).
Before it, one JSON line each for the other single-GPU configs (skipped with --only-headline or an explicit
--workload): cfg3 = configs[2] (the same volume through compute_Sv -> remove_background_noise -> compute_MVBS), cfg4 =
configs[3] (EK80 broadband, 2 x 200 000 x 8192 x 4 sectors: pulse compression + Sv + MVBS), cfg5 = configs[4]'s
32.8 G-sample volume on one GPU as resident tiles.  Every line carries `roofline` (HIP-event time of the dominant
kernel on torch's stream vs the algorithmic bytes of SURVEY 8d) and its own `cpu_baseline` (the NumPy / SciPy oracle
with the reference's pass structure on a bounded slice, host cores, timed before HIP is initialised).

N > 1.  `python bench.py --gpus N` launches its own N ranks (torch.distributed.run, one per GPU, RCCL); under a
launcher (WORLD_SIZE set) it runs as one rank.  Workload = BASELINE configs[4]: EK60 4 ch x 2 M pings x 4096 range
(32.8 G samples) split by ping_time over the N ranks -- STRONG scaling -- each rank holding its share as resident
tiles.  The ping-time origin sits 10 s off the 20-s bin grid, so EVERY tile and shard edge cuts a time bin: the raw
(sum, count) partials of the cut bins go through the edge-bin all-reduce (sharding.EdgeExchange: one RCCL all-reduce
of a few hundred KB per step) inside the timed region.  The same invocation also times the bin-aligned layout (no
collective at all) and reports it as config.aligned_ms_per_step, so the cost of the exchange is on the line.
value = samples processed by all ranks / max-over-ranks wall time of the K timed steps.
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
# name: (C, pings, S, default timed steps, default warmup) -- steps sized for a timed region of ~2 s
WORKLOADS = {
    "cfg2": (4, 500_000, 2000, 220, 10),
    "cfg3": (4, 500_000, 2000, 70, 4),
    "cfg4": (2, 200_000, 8192, 45, 3),
    "cfg4small": (2, 20_000, 8192, 100, 5),
    "cfg5": (4, 2_000_000, 4096, 26, 2),
    "small": (4, 20_000, 2000, 200, 10),
}
TILE_PINGS = 250_000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--ss-every", type=int, default=2000,
                    help="EK60 volumes: the sound speed recorded with the pings changes every this many pings "
                         "(1: at every ping -- no two pings share a range vector)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="one workload only (default: N=1 -> cfg3, cfg4, cfg5 lines then the cfg2 headline; N>1 -> cfg5)")
    ap.add_argument("--only-headline", action="store_true", help="N=1: skip the cfg3 / cfg4 / cfg5 lines")
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--input", default="float32", choices=["float32", "int16"],
                    help="cfg2: float32 = backscatter_r as echopype's converter stores it (the drop-in boundary); "
                         "int16 = the instrument's own samples + ping lengths (SURVEY 8f row 4, 2 B/sample in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain-outputs", default="all", choices=["all", "corrected"],
                    help="cfg3: full-size arrays written -- all = Sv + Sv_noise + Sv_corrected (what "
                         "remove_background_noise adds, SURVEY 8d line E: 32 B/sample fp64); corrected = without Sv_noise")
    ap.add_argument("--pings-total", type=int, default=None, help="N>1 / cfg5: total pings (default 2 000 000)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL; gloo only for dry runs of the N>1 logic)")
    ap.add_argument("--single-device", action="store_true",
                    help="dry run: every rank uses cuda:0 (with --backend gloo)")
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


def _time_runs(run, budget_s=20.0, max_runs=5):
    run()
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < max_runs and (not ts or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def _cpu_worker(a):
    C, P, S, seed = a
    from echopype_amd import synth

    _oracle_ek60(synth.ek60_numpy(C, P, S, seed=20260501 + seed), False)
    return 0


def cpu_baseline_ek60(chain=False, multicore=False):
    """Oracle chain (reference pass structure, NumPy fp64, one core) on BASELINE configs[0]'s shape: EK60 CW
    2 channels x 10 000 pings x 1000 range."""
    from echopype_amd import synth

    C, P, S = 2, 10_000, 1000
    d = synth.ek60_numpy(C, P, S)
    med, n_runs = _time_runs(lambda: _oracle_ek60(d, chain))
    n = C * P * S
    what = "compute_Sv + remove_background_noise(20 x 50, 3 dB) + compute_MVBS" if chain else "compute_Sv + compute_MVBS"
    out = {"value": n / med, "unit": "range-samples/s", "cores": 1, "kind": "port",
           "sample": f"BASELINE configs[0] shape: EK60 {C}ch x {P} pings x {S} range, {what}(20s x 1m), NumPy fp64 "
                     f"oracle (reference pass structure), median of {n_runs} runs, host has {os.cpu_count()} cores"}
    if multicore:  # what dask chunk-parallelism over ping_time could reach at best: the same slice in N processes
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

    med, n_runs = _time_runs(run, budget_s=15.0, max_runs=3)
    return {"value": C * P * S / med, "unit": "range-samples/s", "cores": 1, "kind": "port",
            "sample": f"EK80 BB {C}ch x {P} pings x {S} samples x {B} sectors, {reps[0].size}-tap replicas: "
                      f"scipy.signal.convolve per (ping, sector) + sector mean + Sv (oracle/ek80.py), median of "
                      f"{n_runs} runs, host has {os.cpu_count()} cores"}


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
        tj = json.load(open(path))
        e = tj.get(key)
        if not e:
            return None, "no PMC measurement of this workload under profiles/"
        if e.get("csrc_sha16") != csrc_hash():
            return None, f"stale: PMC measurement in profiles/hbm_traffic.json was taken at csrc {e.get('csrc_sha16')}"
        return e["bytes_per_launch"], (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 correction), "
                                       f"{e.get('source', 'profiles/')}, csrc {e['csrc_sha16']}")
    except Exception as ex:  # noqa: BLE001
        return None, f"unreadable profiles/hbm_traffic.json: {ex!r}"


class Ctx:
    """What every run function needs."""

    def __init__(self, args, world, rank):
        import torch
        import torch.distributed as dist
        from echopype_amd import ops, sharding, synth

        self.args, self.world, self.rank = args, world, rank
        self.torch, self.dist, self.ops, self.sharding, self.synth = torch, dist, ops, sharding, synth
        self.dt = torch.float64 if args.dtype == "float64" else torch.float32

    def sync(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def timed(self, step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks.  step(timer | None)."""
        timers = [self.ops.Timer() for _ in range(steps)]
        for _ in range(warmup):
            step(None)
        self.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(timers[i])
        self.sync()
        elapsed = time.perf_counter() - t0
        kernel_ms = float(np.mean([tm.elapsed_ms() for tm in timers]))
        t = self.torch.tensor([elapsed], dtype=self.torch.float64)
        if self.world > 1:
            t = t.to(self.sharding._comm_device())
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item()), kernel_ms

    def steps(self, workload):
        a = self.args
        # (strong scaling: a rank's step shrinks with the world size; the default keeps the timed region near 2 s)
        scale = self.world if workload == "cfg5" else 1
        return (a.steps if a.steps is not None else WORKLOADS[workload][3] * scale,
                a.warmup if a.warmup is not None else WORKLOADS[workload][4])

    def free(self):
        import gc

        gc.collect()
        self.torch.cuda.empty_cache()


def line(ctx, *, value, steps, warmup, elapsed, scaling, workload, config, roofline, metric=METRIC, cpu=None):
    out = {"metric": metric, "value": value, "unit": "range-samples/s", "n_gpus": ctx.world, "steps": steps,
           "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": scaling,
           "vs_baseline": None, "dtype": "f64" if ctx.args.dtype == "float64" else "f32", "data": "synthetic",
           "config": {"workload": workload, **config,
                      "synthetic_data": "10 % of pings short and NaN-padded; the recorded sound speed follows a slow "
                                        f"drift (EK60: a new value every {ctx.args.ss_every} pings, EK80: every ping)"},
           "roofline": roofline}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def roofline(kernel, kernel_ms, bytes_per_launch, bps, traffic_key=None, **extra):
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic, src = measured_traffic(traffic_key) if traffic_key else (None, "not measured for this workload")
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_sample": bps,
            **extra}


# ---------------------------------------------------------------------------------------- EK60: cfg2 / cfg3
def _coef(ctx, d):
    return ctx.ops.power_coef_ek(
        d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"], d["sound_speed_indicative"],
        d["absorption_indicative"], d["gain_correction"], d["sa_correction"], d["equivalent_beam_angle"],
        d["frequency_nominal"], d["tau0"], pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)


def api_ms(ctx, C, P, S, raw):
    """The same step through the drop-in Dataset API on the resident samples: (compute_Sv_MVBS(echodata), compute_Sv(echodata),
    compute_MVBS(ds_Sv)) in ms."""
    import logging

    import echopype_amd as ep

    d = ctx.synth.ek60_numpy(C, 4, 8)
    for k, v in list(d.items()):
        if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == (C, 4):
            d[k] = np.repeat(v[:, :1], P, axis=1)
    p = np.arange(P)
    d["sound_speed_indicative"] = np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1))
    d["backscatter_r"] = ep.DeviceArray(raw)
    d["ping_time"] = ctx.synth.T0 + (p * 1_000_000_000).astype("timedelta64[ns]")
    ed = ep.echodata.from_ek60_arrays(d).to_device()  # samples AND per-ping parameters resident in HBM
    logging.disable(logging.WARNING)
    dtype = ctx.args.dtype

    def med(f):
        r = f()
        ctx.torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            del r
            ctx.torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = f()
            ctx.torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, r

    try:
        one, r = med(lambda: ep.compute_Sv_MVBS(ed, range_bin="1m", ping_time_bin="20s", dtype=dtype))
        del r
        # the reference's own two calls
        sv_ms, ds = med(lambda: ep.calibrate.compute_Sv(ed, dtype=dtype))
        mv_ms, r = med(lambda: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s"))
        del r, ds
    finally:
        logging.disable(logging.NOTSET)
    return one, sv_ms, mv_ms


def run_ek60(ctx, name, cpu):
    args, torch, ops, sharding = ctx.args, ctx.torch, ctx.ops, ctx.sharding
    C, P, S = WORKLOADS[name][:3]
    chain = name == "cfg3"
    i16 = args.input == "int16" and not chain
    dt = ctx.dt
    d = (ctx.synth.ek60_device_i16 if i16 else ctx.synth.ek60_device)(C, P, S, seed=20260501, ss_every=args.ss_every)
    ns = d["ping_time_ns"]
    bin_ns = 20_000_000_000
    e0, _ = sharding.global_time_grid(ns.cpu().numpy(), bin_ns)
    n_t = P // 20
    r_max = float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2)  # analytic for this recipe
    n_r = len(np.arange(0, r_max + 1.0, 1.0)) - 1
    sv = torch.empty((C, P, S), dtype=dt, device="cuda") if not chain else None
    mvbs = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
    d["tau0"] = d["transmit_duration_nominal"][:, 0].contiguous()  # EK60 tau_eff = ping 0

    def step(timer):
        coef = _coef(ctx, d)
        bs = ops.time_bin_offsets(ns, e0, bin_ns, n_t)
        if timer is not None:
            timer.start()
        if chain:  # noise blocks of 20 pings x 50 samples, SNR 3 dB
            a2 = coef[..., 4].contiguous()
            _, _, nz = ops.sv_noise_fused(d["backscatter_r"], coef, a2, 20, 50, dtype=dt)
            ops.sv_denoise_mvbs(d["backscatter_r"], coef, a2, nz, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt,
                                want_noise=args.chain_outputs == "all")
        elif i16:
            ops.sv_mvbs_fused_i16(d["raw_i16"], d["n_valid"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mvbs)
        else:
            ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mvbs)
        if timer is not None:
            timer.stop()

    steps, warmup = ctx.steps(name)
    elapsed, kernel_ms = ctx.timed(step, steps, warmup)
    n = C * P * S
    bps = BYTES_PER_SAMPLE[args.dtype] - (2 if i16 else 0)
    if chain:  # two sweeps over the 4-B input + Sv, Sv_corrected (+ Sv_noise) out: SURVEY 8d line E = 32 / 20 B
        bps = 2 * BYTES_PER_SAMPLE[args.dtype] + ((BYTES_PER_SAMPLE[args.dtype] - 4) if args.chain_outputs == "all" else 0)
    key = f"{name}:{args.dtype}" + (":int16" if i16 else "") + (":corrected" if chain and args.chain_outputs != "all" else "")
    cfg = {"pings_total": P, "sharding": "one GPU", "collective": "none"}
    if not chain and not i16:
        del sv, mvbs
        ctx.free()
        cfg["api_ms_per_step"], sv_ms, mv_ms = api_ms(ctx, C, P, S, d["backscatter_r"])
        cfg["api_two_calls_ms"] = {"compute_Sv": sv_ms, "compute_MVBS": mv_ms}
        cfg["api_note"] = ("api_ms_per_step: echopype_amd.compute_Sv_MVBS(echodata) through the Dataset API on the same "
                           "volume, echodata resident in HBM (EchoData.to_device: samples and per-ping parameters): "
                           "parameter selection + kernels + Dataset assembly, median of 5 calls; api_two_calls_ms: the "
                           "reference's own sequence calibrate.compute_Sv(echodata) then commongrid.compute_MVBS(ds_Sv) "
                           "(echo_range stays lazy, binned through its coefficient rows)")
    what = ("fused compute_Sv -> compute_MVBS (20 s x 1 m), Sv + MVBS written" if not chain else
            "two-pass compute_Sv -> remove_background_noise (20 x 50, 3 dB) -> compute_MVBS of Sv_corrected (20 s x 1 m), "
            "Sv + " + ("Sv_noise + " if args.chain_outputs == "all" else "") + "Sv_corrected + MVBS written")
    return line(ctx, value=n * steps / elapsed, steps=steps, warmup=warmup, elapsed=elapsed, scaling="weak",
                metric=METRIC + (" with remove_background_noise" if chain else ""),
                workload=f"EK60 CW {C}ch x {P} pings x {S} range ({name}), {what}" + (", int16 instrument samples in" if i16 else ""),
                config=cfg, cpu=cpu,
                roofline=roofline("epa_chain::sv_noise_fast_kernel + sv_denoise_mvbs_uniform_kernel (+ sv_denoise_mvbs_fast_kernel for time bins whose pings differ)" if chain
                                  else "epa_fused::fused_sv_mvbs_kernel", kernel_ms, n * bps, bps, traffic_key=key))


# ---------------------------------------------------------------------------------------- EK80 BB: cfg4
def run_ek80(ctx, name, cpu):
    """EK80 BB complex -> pulse compression + Sv (epa_sv_complex_fft) -> MVBS.  The per-(channel, ping) parameter rows
    and the replicas are assembled once by the drop-in's own calibrator (host, O(C*P)); one step = the two kernels over
    the resident planes."""
    import echopype_amd as ep
    from echopype_amd import _lib
    from echopype_amd.calibrate.api import CALIBRATOR

    args, torch, ops, sharding, synth = ctx.args, ctx.torch, ctx.ops, ctx.sharding, ctx.synth
    C, P, S = WORKLOADS[name][:3]
    B = 4
    d = synth.ek80_numpy(C, 4, 64, B)  # parameters only; the sample planes are generated on the device
    g = torch.Generator(device="cuda")
    g.manual_seed(20260504)
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
    pidx = np.arange(P)
    ping_time = synth.T0 + (pidx * 1_000_000_000).astype("timedelta64[ns]")
    d.update(backscatter_r=ep.DeviceArray(re), backscatter_i=ep.DeviceArray(im), sample_interval=np.full((C, P), 8e-6),
             sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * pidx / 1e5), (C, 1)), ping_time=ping_time)
    ed = ep.echodata.from_ek80_arrays(d, synth.ek80_filters())
    cal = CALIBRATOR["EK80"](ed, env_params=None, cal_params=None, ecs_file=None, waveform_mode="BB",
                             encode_mode="complex", dtype=args.dtype, device=None)
    k, _ = cal._complex_inputs("Sv")
    ns = torch.from_numpy(ping_time.astype(np.int64)).cuda()
    bin_ns = 20_000_000_000
    e0, _ = sharding.global_time_grid(ns.cpu().numpy(), bin_ns)
    n_t = P // 20
    range_bin = 0.1
    rmax = float((S - 1) * 8e-6 * 1500.5 / 2)
    n_r = len(np.arange(0, rmax + range_bin, range_bin)) - 1
    rows = torch.zeros((C, P, _lib.NCOEF), dtype=torch.float64, device="cuda")
    rows[..., _lib.CF_RA] = k["ccoef"][..., _lib.CC_RA]
    rows[..., _lib.CF_RB] = k["ccoef"][..., _lib.CC_RB]

    def step(timer):
        bs = ops.time_bin_offsets(ns, e0, bin_ns, n_t)
        if timer is not None:
            timer.start()
        res = ops.sv_complex(k["re"], k["im"], k["ccoef"], replica=k["replica"], replica_off=k["replica_off"],
                             max_taps=k["max_taps"], dtype=ctx.dt, want_range=False)
        if timer is not None:
            timer.stop()
        ops.mvbs(res["out"], bs, n_t, range_bin, n_r, coef=rows)

    steps, warmup = ctx.steps(name)
    elapsed, kernel_ms = ctx.timed(step, steps, warmup)
    n = C * P * S
    bps = B * 8 + (8 if args.dtype == "float64" else 4)  # SURVEY 8d line F: complex64 sectors in, Sv out
    taps = int(k["max_taps"])
    return line(ctx, value=n * steps / elapsed, steps=steps, warmup=warmup, elapsed=elapsed, scaling="weak", cpu=cpu,
                workload=f"EK80 BB complex {C}ch x {P} pings x {S} samples x {B} sectors ({name}), float32 planes "
                         f"resident, {taps}-tap replica: pulse compression + Sv, then MVBS (20 s x {range_bin} m); a "
                         "sample = one (channel, ping, range_sample) output",
                config={"pings_total": P, "sharding": "one GPU", "collective": "none",
                        "fft_dtype": "complex128" if args.dtype == "float64" else "complex64"},
                roofline=roofline("sv_complex_fft_kernel", kernel_ms, n * bps, bps, traffic_key=f"{name}:{args.dtype}",
                                  direct_form_tflops=8.0 * taps * n / (kernel_ms * 1e-3) / 1e12,
                                  note="in-place LDS FFT (DIF / DIT, 6 LDS round trips per 2048-sample tile); the "
                                       "transform runs in the output's precision.  Instruction counters of this kernel "
                                       "(profiles/r02_pmc_hot.txt, a replayed measurement): 241 (complex128) / 232 "
                                       "(complex64) VALU wavefront instructions per output sample = 20 / 19 ms of pure "
                                       "VALU issue at this volume -- the kernel meets its instruction-issue bound "
                                       "before the HBM one"))


# ---------------------------------------------------------------------------------------- cfg5: tiles, N >= 1
def run_cfg5(ctx, cpu):
    """BASELINE configs[4]: 4 x 2 M x 4096 split by ping_time over the ranks (STRONG scaling), each rank's share as
    resident tiles of <= 250 000 pings.  N = 1: Sv of every tile goes to one reused buffer (131 GB in + 262 GB out does
    not fit 288 GB otherwise) and tile edges sit on bin edges.  N > 1: two layouts are timed, see the module docstring."""
    args, torch, ops, sharding, synth, world, rank = ctx.args, ctx.torch, ctx.ops, ctx.sharding, ctx.synth, ctx.world, ctx.rank
    C, _, S = WORKLOADS["cfg5"][:3]
    P_total = args.pings_total or WORKLOADS["cfg5"][1]
    dt = ctx.dt
    bin_ns = 20_000_000_000
    p0, p1 = sharding.shard_bounds(P_total, world, rank, align=20)
    tile_p = min(TILE_PINGS, max(20, p1 - p0))
    spans = [(a, min(p1, a + tile_p)) for a in range(p0, p1, tile_p)]
    tiles = []
    for a, b in spans:
        d = synth.ek60_device(C, b - a, S, seed=20260505 + a // 20, ping0=a, ss_every=args.ss_every)
        d["tau0"] = torch.full((C,), 1.024e-3, dtype=torch.float64, device="cuda")  # ping 0 of the WHOLE file
        tiles.append(d)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
    esz = 8 if args.dtype == "float64" else 4
    free_b = torch.cuda.mem_get_info()[0]
    keep_all = sum((b - a) for a, b in spans) * C * S * esz < free_b - (8 << 30)  # Sv of every tile resident?
    if keep_all:
        sv = [torch.empty((C, b - a, S), dtype=dt, device="cuda") for a, b in spans]
    else:  # one buffer, every tile writes its Sv through a contiguous view of it
        buf = torch.empty((C, tile_p, S), dtype=dt, device="cuda")
        sv = [buf.view(-1)[:C * (b - a) * S].view(C, b - a, S) for a, b in spans]
    steps, warmup = ctx.steps("cfg5")

    def layout(offset_ns):
        """Time grid + exchange plan for ping times shifted by ``offset_ns`` against the 20-s grid."""
        info = []
        ts = [d["ping_time_ns"] + offset_ns for d in tiles]
        ends = np.array([int(x) for t in ts for x in (t[0], t[-1])], dtype=np.int64)
        e0, _ = sharding.global_time_grid(ends, bin_ns)  # ONE grid for every tile of every rank
        for t in ts:
            f, l = sharding.local_bin_span(t[[0, -1]].cpu().numpy(), e0, bin_ns)
            info.append((t, e0 + f * bin_ns, f, l, l - f + 1))
        plan = sharding.EdgeExchange([(f, l) for _, _, f, l, _ in info], C, n_r, "cuda")
        mv = [torch.empty((C, n, n_r), dtype=dt, device="cuda") for *_, n in info]
        return info, plan, mv

    def make_step(info, plan, mv):
        def step(timer):
            rows = {}
            for i, d in enumerate(tiles):
                t, e0l, f, l, n_t = info[i]
                coef = _coef(ctx, d)
                bs = ops.time_bin_offsets(t, e0l, bin_ns, n_t)
                if timer is not None and i == 0:
                    timer.start()
                res = ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv[i],
                                        mvbs_out=mv[i], want_partials=plan.shared)
                if timer is not None and i == 0:
                    timer.stop()
                if plan.shared:
                    for w, r in sharding.mvbs_edge_rows(res["sum"], res["cnt"]).items():
                        rows[(i, w)] = r
            if plan.shared:  # ONE all-reduce for every cut bin of every tile of every rank, then finalise the owners' rows
                tot = plan.merge(rows)
                for k, w, _, owner in plan.edges:
                    if owner:
                        s, c = tot[(k, w)]
                        mv[k][:, 0 if w == 0 else -1] = ops.mvbs_finalize(s.to(dt).contiguous(), c.to(torch.int32).contiguous())
        return step

    # (a) bin-aligned layout: no bin is shared, no collective
    info_a, plan_a, mv_a = layout(0)
    assert not plan_a.shared
    el_a, km_a = ctx.timed(make_step(info_a, plan_a, mv_a), steps, warmup)
    out_cfg = {"pings_total": P_total, "sharding": f"ping_time x{world}, {len(spans)} resident tile(s) of <= {tile_p} pings per rank",
               "sv_resident": "every tile" if keep_all else "one reused tile buffer"}
    elapsed, kernel_ms, coll = el_a, km_a, "none (tile and shard edges on bin edges)"
    if world > 1 or len(spans) > 1:
        del mv_a
        # (b) ping times 10 s off the grid: every tile / shard edge cuts a bin -> edge-bin all-reduce on the timed path
        info_b, plan_b, mv_b = layout(10_000_000_000)
        assert plan_b.shared
        el_b, km_b = ctx.timed(make_step(info_b, plan_b, mv_b), steps, warmup)
        out_cfg.update(aligned_ms_per_step=el_a / steps * 1e3, straddle_ms_per_step=el_b / steps * 1e3,
                       collective_ms_per_step=(el_b - el_a) / steps * 1e3,
                       edge_bins_per_step=len(plan_b.edges), allreduce_bytes=int(plan_b._buf.numel() * 8))
        if world > 1:  # the headline of an N > 1 run is the layout WITH the exchange
            elapsed, kernel_ms = el_b, km_b
            coll = (f"edge-bin all-reduce ({args.backend}): one all_reduce(SUM) of {plan_b._buf.numel() * 8} B per step "
                    "for the time bins cut by tile / shard edges")
    out_cfg["collective"] = coll
    n_first = C * (spans[0][1] - spans[0][0]) * S if spans else 0
    bps = BYTES_PER_SAMPLE[args.dtype]
    if rank != 0:
        return None
    return line(ctx, value=C * P_total * S * steps / elapsed, steps=steps, warmup=warmup, elapsed=elapsed,
                scaling="strong", cpu=cpu,
                workload=f"EK60 CW {C}ch x {P_total} pings x {S} range TOTAL (cfg5 = BASELINE configs[4]), split by "
                         f"ping_time over {world} rank(s), fused compute_Sv -> compute_MVBS (20 s x 1 m), Sv + MVBS written",
                config=out_cfg,
                roofline=roofline("epa_fused::fused_sv_mvbs_kernel", kernel_ms, n_first * bps, bps,
                                  note="kernel_ms = launch over one tile (rank 0, first tile)"))


# ---------------------------------------------------------------------------------------- main
def relaunch(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU and pass its single JSON line through."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


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
        todo = [args.workload]
    else:
        todo = (["cfg2"] if args.only_headline else ["cfg3", "cfg4", "cfg4:f32", "cfg5", "cfg2"])
    # CPU baselines first: they fork worker processes, which must happen before HIP is initialised
    cpu = {}
    if world == 1 and not args.no_cpu_baseline:
        for w in todo:
            kind = {"cfg3": "chain", "cfg4": "bb", "cfg4small": "bb"}.get(w.partition(":")[0], "ek60")
            if kind not in cpu:
                cpu[kind] = (cpu_baseline_bb() if kind == "bb" else
                             cpu_baseline_ek60(chain=kind == "chain", multicore=kind == "ek60"))
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    ctx = Ctx(args, world, rank)
    for w in todo:
        w, _, other_dtype = w.partition(":")  # "cfg4:f32": the same workload with float32 output (complex64 transform)
        kind = {"cfg3": "chain", "cfg4": "bb", "cfg4small": "bb"}.get(w, "ek60")
        asked = args.dtype
        if other_dtype:
            args.dtype = {"f32": "float32", "f64": "float64"}[other_dtype]
            ctx.dt = torch.float64 if args.dtype == "float64" else torch.float32
        if w == "cfg5":
            out = run_cfg5(ctx, cpu.get(kind))
        elif w.startswith("cfg4"):
            out = run_ek80(ctx, w, cpu.get(kind))
        else:
            out = run_ek60(ctx, w, cpu.get(kind))
        if rank == 0 and out is not None:
            print(json.dumps(out), flush=True)
        ctx.free()
        args.dtype = asked
        ctx.dt = torch.float64 if asked == "float64" else torch.float32
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
