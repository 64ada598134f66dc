"""The one-JSON-line-per-workload contract of bench.py: the committed lines of the final run (CPU) and a live headline
line (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOP = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
       "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def _check_line(d, want_cpu):
    for k, t in TOP.items():
        assert k in d, k
        assert isinstance(d[k], (int, float)) if t is float else isinstance(d[k], t), (k, d[k])
    assert "vs_baseline" in d and d["vs_baseline"] is None        # BASELINE.md holds no published number for this metric
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and d["scaling"] in ("weak", "strong")
    assert d["dtype"] in ("f64", "f32") and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and 0 < r["frac"] < 1  # (seven significant digits are printed)
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001            # the dominant kernel is part of the step
    assert r["traffic"] is None or r["traffic"] > 0
    if want_cpu:
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


LINES = os.path.join(ROOT, "profiles", "r06_bench_default.jsonl")


def test_committed_bench_lines_keep_the_contract():
    if not os.path.exists(LINES):
        pytest.skip("no committed default run of this round yet")
    raw = [x.strip() for x in open(LINES) if x.strip()]
    assert all(len(x) < 2000 for x in raw), [len(x) for x in raw]                     # a line fits the driver's tail
    lines = [json.loads(x) for x in raw]
    assert len(lines) == 25
    for d in lines:
        _check_line(d, want_cpu=True)
        assert d["n_gpus"] == 1 and d["config"]["passes_per_step"] >= 1
        if d["config"]["workload"].startswith("next:"):  # (a SURVEY 8f row: its own metric string)
            assert "SURVEY 8f" in d["metric"]
        n = d["config"]["samples_per_step"]
        assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6    # value = samples per step / step time
    head = lines[-1]                                              # the headline (BASELINE.json's metric) comes last
    assert head["config"]["workload"].startswith("cfg5") and head["metric"].startswith("range-samples/sec")
    assert head["scaling"] == "strong" and head["config"]["samples_per_step"] == 4 * 2_000_000 * 4096 * head["config"]["passes_per_step"]
    # the headline goes through the product entry points (echopype_amd.pipeline deals the tiles to two streams); the
    # ops-level harness on the same tiles rides beside it -- on the one-stream line just before when the headline had to
    # drop optional keys to fit (bench.fit, config.dropped)
    one = lines[-2]
    assert "compute_Sv -> compute_MVBS" in head["config"]["route"]
    ops_cfg = head["config"] if "ops_level_ms_per_pass" in head["config"] else one["config"]
    assert ops_cfg["ops_level_ms_per_pass"] > 0 and ops_cfg["allreduce_bytes"] > 0 and ops_cfg["ops_level_edge_bins"] == 14
    assert head["config"]["ms_per_pass"] < 1.05 * ops_cfg["ops_level_ms_per_pass"]       # within 5 % of the kernels alone
    assert head["config"].get("dropped", 0) <= 6 and "collective" in head["config"]
    assert head["config"]["ranks"]["world_size"] == 1 and len(head["config"]["ranks"]["devices"]) == 1
    assert head["roofline"]["frac"] >= 0.60
    no_traffic = [d["config"]["workload"][:12] for d in lines if d["roofline"]["traffic"] is None]
    assert no_traffic == ["api:pcie: EK"], no_traffic            # (bound by PCIe: no HBM figure is claimed for it)
    also = {k for k in head["config"] if k.startswith("also_")}   # the other lines' figures, one flat string per family
    assert also == {"also_cfg3", "also_cfg2", "also_api", "also_pcie", "also_cfg4", "also_next", "also_cfg5", "also_unit"}
    # the tiles go to two streams: the roofline's duration is the wall time a launch costs, the HIP-event bracket of one
    # launch (sharing the GPU with its neighbour) rides beside it; the one-stream line comes just before the headline
    assert head["config"]["tile_streams"] == 2 and head["roofline"]["kernel_ms_each"] > head["roofline"]["kernel_ms"]
    assert abs(head["roofline"]["kernel_ms"] * 8 - head["config"]["ms_per_pass"]) < 1e-3 * head["config"]["ms_per_pass"]
    assert one["config"]["workload"] == head["config"]["workload"] and "tile_streams" not in one["config"]
    assert "kernel_ms_each" not in one["roofline"] and head["config"]["also_cfg5"].startswith("one ")
    assert head["config"]["also_next"].count(";") == 5 and "int16" in head["config"]["also_cfg2"]  # the SURVEY 8f rows
    assert all(k in head["config"]["also_next"] for k in ("depth ", "depthw ", "masks ", "masks2000 ", "masksidx ", "nasc "))
    pcie = [d for d in lines if d["config"]["workload"].startswith("api:pcie")]                   # PCIe-inclusive, never `value`
    assert len(pcie) == 1 and pcie[0]["value"] < 0.1 * head["value"] and head["config"]["also_pcie"]
    cb = head["cpu_baseline"]                                     # one core + every core the box grants, bounded by time
    assert cb["multicore_cores"] >= cb["cores"] == 1 and "multicore_sample" in cb
    assert "sv " in head["config"]["also_cfg2"] and "sv32 " in head["config"]["also_cfg2"]             # K1 alone (configs[1])
    assert head["config"]["host_ms_per_call"] > 0
    assert all(len(head["config"][k]) <= 130 for k in also)
    assert head["config"]["also_cfg3"].count(";") == 2 and "ss2000" in head["config"]["also_cfg3"]
    by = {d["config"]["workload"].split(":")[0] + ":" + d["dtype"] for d in lines}
    assert {"cfg2:f64", "cfg2:f32", "cfg3:f64", "cfg3:f32", "cfg4:f64", "cfg4:f32", "cfg5:f64", "api:f64", "next:f64"} <= by
    # the chain through the reference's THREE calls runs at the chain kernels' own speed (ops-level: the cfg3 line)
    chain = [d for d in lines if d["config"]["workload"].startswith("api:chain")]
    cfg3 = [d for d in lines if d["config"]["workload"].startswith("cfg3") and d["dtype"] == "f64"]
    assert len(chain) == 1 and "remove_background_noise" in chain[0]["config"]["workload"]
    # (both run the recipe with a new sound speed at every ping; the every-2000-pings variant is a little faster)
    assert chain[0]["config"]["ms_per_pass"] < 1.08 * max(d["config"]["ms_per_pass"] for d in cfg3)
    assert "chain" in head["config"]["also_api"]


def test_fit_bounds_a_line_by_dropping_optional_keys_in_order():
    """bench.fit: the printed line stays under the driver's 2000-character tail whatever the other lines' figures add to
    the headline; only keys of DROP_ORDER go, in that order, and the contract's keys never do."""
    sys.path.insert(0, ROOT)
    import bench

    raw = [x.strip() for x in open(LINES) if x.strip()] if os.path.exists(LINES) else []
    if not raw:
        pytest.skip("no committed default run of this round yet")
    head = json.loads(raw[-1])
    head["config"].pop("dropped", None)
    assert json.loads(bench.fit(dict(head), limit=10_000)) == bench.compact(head)       # fits: nothing is dropped
    fat = json.loads(raw[-1])
    fat["config"]["ops_level_edge_bins"], fat["config"]["allreduce_bytes"] = 14, 805888
    fat["config"]["also_next"] = fat["config"]["also_next"] + "; " + "x" * 400
    txt = bench.fit(fat)
    out = json.loads(txt)
    assert len(txt) <= 1950 and out["config"]["dropped"] >= 2
    gone = [k for sec, k in bench.DROP_ORDER if k not in out.get(sec, {})]
    kept = [k for sec, k in bench.DROP_ORDER if k in out.get(sec, {})]
    order = [k for _, k in bench.DROP_ORDER]
    assert not kept or not gone or max(order.index(k) for k in gone) < min(order.index(k) for k in kept) or \
        all(k not in fat.get(sec, {}) for sec, k in bench.DROP_ORDER if k in gone and order.index(k) > min(order.index(q) for q in kept))
    for k in TOP:
        assert k in out
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"])


@pytest.mark.gpu
def test_live_headline_line():
    """The N = 1 headline at a reduced ping count (the full volume is the bench's own business and
    tests/test_gpu_fullsize_cfg5.py's): same code path, tiles, straddling layout and exchange."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only-headline", "--no-cpu-baseline",
                          "--steps", "3", "--warmup", "1", "--pings-total", "400000"], capture_output=True, text=True,
                         cwd=ROOT, check=True)
    lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_line(d, want_cpu=False)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert d["config"]["workload"].startswith("cfg5") and d["config"]["allreduce_bytes"] > 0
    assert "compute_Sv -> compute_MVBS" in d["config"]["route"] and d["config"]["mvbs_shape_last_tile"][0] == 4
    assert d["config"]["ranks"]["world_size"] == 1 and d["config"]["ranks"]["devices"] == ["0:cuda0"]
    assert len(lines[0]) < 1800                                   # (room for eight ranks' devices under the driver's 2000)


@pytest.mark.gpu
def test_gloo_two_ranks_print_the_same_workload_with_a_cpu_baseline():
    """`--gpus 2 --backend gloo --single-device` (two processes on cuda:0) and `--gpus 1` print the same workload up to
    the split, each with roofline + cpu_baseline (the N > 1 line used to come without one)."""
    common = ["--steps", "2", "--warmup", "1", "--pings-total", "200000", "--workload", "cfg5"]
    outs = []
    for extra in ([], ["--gpus", "2", "--backend", "gloo", "--single-device"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, *extra], capture_output=True,
                           text=True, cwd=ROOT, check=True)
        lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
        assert len(lines) == 1
        outs.append(json.loads(lines[0]))
    one, two = outs
    _check_line(one, want_cpu=True)
    _check_line(two, want_cpu=True)
    assert one["config"]["workload"] == two["config"]["workload"] and one["scaling"] == two["scaling"] == "strong"
    assert (one["n_gpus"], two["n_gpus"]) == (1, 2)
    assert one["config"]["samples_per_step"] == two["config"]["samples_per_step"]
    assert one["config"]["tiles"].split(" over ")[0] == two["config"]["tiles"].split(" over ")[0]
    # N = 1: the reference's two calls per tile; N > 1: the sharded entry point, every rank and its device on the line
    assert "sharding.compute_Sv_MVBS" in two["config"]["route"] and "compute_Sv -> compute_MVBS" in one["config"]["route"]
    r = two["config"]["ranks"]
    assert r["world_size"] == 2 and r["backend"] == "gloo" and r["devices"] == ["0:cuda0", "1:cuda0"]
    assert one["config"]["mvbs_shape_last_tile"][2] == two["config"]["mvbs_shape_last_tile"][2]


@pytest.mark.gpu
def test_sharded_route_on_a_one_rank_rccl_group_prints_its_host_cost():
    """`--sharded-at-1`: the SHARDED entry points per tile on a one-rank RCCL group (every collective of the N > 1 route
    executes, as an identity) -- the line carries the host time per call next to the N = 1 figures."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only-headline", "--no-cpu-baseline", "--steps", "2",
                        "--warmup", "1", "--pings-total", "200000", "--sharded-at-1"], capture_output=True, text=True,
                       cwd=ROOT, check=True)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_line(d, want_cpu=False)
    assert d["n_gpus"] == 1 and "sharding.compute_Sv_MVBS" in d["config"]["route"]
    assert 0 < d["config"]["host_ms_per_call"] < 50 and d["config"]["allreduce_bytes"] > 0
