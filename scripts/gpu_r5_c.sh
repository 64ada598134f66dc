#!/bin/bash
# round 5: GPU suite on the tree (fused lean path, sharded deferred routes, file scalars), the new bench lines, the host
# cost of the sharded route on one GPU (one-rank RCCL group; gloo control-plane latencies at 2/4/8 ranks)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=40 > $O/tests.txt 2>&1; echo "tests rc $?"
tail -30 $O/tests.txt
for wl in next:depth next:masks next:nasc cfg2:int16; do
  python bench.py --workload $wl --no-cpu-baseline --steps 5 --warmup 2 2>$O/err_$(echo $wl | tr ':' '_').txt | tail -1 >> $O/bench_next.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r5c/bench_next.jsonl"):
    if not l.strip(): continue
    d = json.loads(l)
    print(d["config"]["workload"][:60], "| %.2f G/s  %.2f ms/pass  region %.2f ms  frac %.3f" % (d["value"] / 1e9, d["config"]["ms_per_pass"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
tail -3 $O/err_*.txt
# the headline: N = 1 product route, and the SHARDED route on a one-rank RCCL group (host cost of the N > 1 path)
python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 2>$O/err_head.txt | tail -1 > $O/bench_head.json
python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 --sharded-at-1 2>$O/err_head_sharded.txt | tail -1 > $O/bench_head_sharded_at_1.json
python bench.py --gpus 2 --backend gloo --single-device --no-cpu-baseline --steps 5 --warmup 2 2>$O/err_gloo2.txt | tail -1 > $O/bench_gloo2_full_tiles.json
python - <<'PY'
import json
for f in ("bench_head", "bench_head_sharded_at_1", "bench_gloo2_full_tiles"):
    try:
        d = json.loads(open(f"gpurun_out/r5c/{f}.json").read())
        c = d["config"]
        print(f, "| %.1f G/s  %.2f ms/pass  ops-level %.2f ms/pass  kernel %.2f ms  frac %.3f  host %.3f ms/call" % (
            d["value"] / 1e9, c["ms_per_pass"], c["ops_level_ms_per_pass"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], c["host_ms_per_call"]))
    except Exception as e:
        print(f, "FAILED", repr(e)[:200])
PY
tail -5 $O/err_head_sharded.txt $O/err_gloo2.txt
python scripts/probe_gloo_latency.py > $O/gloo_latency.txt 2>&1; tail -8 $O/gloo_latency.txt
