#!/bin/bash
# one GPU call: the whole -m gpu suite, the default bench (all lines), a 2-rank single-GPU dry run of the N>1 path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/full; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/tests.txt; cat $O/tests.txt
python bench.py > $O/bench_default.jsonl 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
for l in open("gpurun_out/full/bench_default.jsonl"):
    d = json.loads(l); c = d["config"]
    print(d["config"]["workload"][:60], "| %.1f G/s  %.2f ms/step  kernel %.2f ms  frac %.3f  cpu %.3g" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value", 0)), {k: round(v, 2) for k, v in c.items() if k.endswith("ms_per_step")})
PY
python bench.py --gpus 2 --backend gloo --single-device --pings-total 400000 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -c 400 $O/bench_gloo2.err; cat $O/bench_gloo2.json | cut -c1-1500
