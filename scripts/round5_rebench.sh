#!/bin/bash
# The bench part of scripts/round5_final.sh alone (csrc unchanged since the PMC passes: profiles/hbm_traffic.json stays valid)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final5; mkdir -p $O
rm -f $O/bench_default.jsonl
python bench.py --steps 20 --warmup 3 --out $O/bench_default.jsonl > $O/bench_default.log 2> $O/bench_default.err
python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 --sharded-at-1 2>$O/bench_sharded_at_1.err | tail -1 > $O/bench_sharded_at_1.json
python bench.py --gpus 2 --backend gloo --single-device --no-cpu-baseline --steps 5 --warmup 2 2>$O/bench_gloo2.err | tail -1 > $O/bench_gloo2.json
rm -rf $O/kt; rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_under_rocprof.jsonl 2> $O/bench_under_rocprof.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -delete
python -m pytest tests/test_bench_contract.py -m gpu -q > $O/tests_bench.txt 2>&1; tail -3 $O/tests_bench.txt
python - <<'PY'
import json
for l in open("gpurun_out/final5/bench_default.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:50], d["dtype"], "| %.1f G/s  %.2f ms/pass  kernel %.2f ms  frac %.3f  traffic %s" % (
        d["value"] / 1e9, d["config"]["ms_per_pass"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
        None if d["roofline"]["traffic"] is None else round(d["roofline"]["traffic"] / 1e9, 2)), len(l))
for f in ("bench_sharded_at_1", "bench_gloo2"):
    d = json.loads(open(f"gpurun_out/final5/{f}.json").read()); c = d["config"]
    print(f, "| %.1f G/s  %.2f ms/pass  ops-level %.2f  kernel %.2f ms  frac %.3f  host %.3f ms/call" % (d["value"] / 1e9, c["ms_per_pass"], c["ops_level_ms_per_pass"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], c["host_ms_per_call"]))
PY
