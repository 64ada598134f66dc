#!/bin/bash
# round 6: kernel stats of the index-binned masks line, the 2-rank gloo dry run, smoke()
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6idx; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python bench.py --workload next:masksidx,next:masks2000 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench.jsonl 2> $O/bench.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r6idx/kernel_stats.csv")))
for r in rows[:40]:
    print(r["Name"][:100].ljust(100), r["Calls"], "%.3f ms avg" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
PY
python bench.py --gpus 2 --backend gloo --single-device --no-cpu-baseline --steps 5 --warmup 2 2>$O/bench_gloo2.err | tail -1 > $O/bench_gloo2.json
cut -c1-700 $O/bench_gloo2.json; tail -n 3 $O/bench_gloo2.err | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
