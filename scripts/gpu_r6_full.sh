#!/bin/bash
# round 6: the default bench (every line, with the CPU baselines) + the full GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6full; rm -rf $O; mkdir -p $O
( time timeout 1500 python bench.py --out $O/bench_default.jsonl ) > $O/bench_stdout.txt 2> $O/bench_err.txt
tail -n 3 $O/bench_err.txt
python - <<'PY'
import json
lines = [l for l in open("gpurun_out/r6full/bench_default.jsonl")]
for l in lines:
    d = json.loads(l)
    print(d["config"]["workload"][:46].ljust(48), d["dtype"], "Gs/s %7.1f" % (d["value"]/1e9), "ms/pass %8.3f" % d["config"]["ms_per_pass"], "frac %.3f" % d["roofline"]["frac"], "B/s", d["roofline"]["bytes_per_sample"])
print("headline characters:", len(lines[-1]))
d = json.loads(lines[-1]); print({k: v for k, v in d["config"].items() if k.startswith("also")}); print(d.get("cpu_baseline"))
PY
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1
tail -n 4 $O/tests_all.txt
