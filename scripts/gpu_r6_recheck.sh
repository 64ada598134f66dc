#!/bin/bash
# round 6, after the final run: the index-binned masks line (with its PMC traffic, merged into profiles/hbm_traffic.json),
# the default bench again (line length, bounded multicore baseline), the 2-rank gloo dry run on one stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/recheck6; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  n=fetch; [ $c = "FETCH_SIZE" ] || n=write
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/${n}_next_masksidx -o p --output-format csv -- python bench.py --workload next:masksidx --no-cpu-baseline --steps 1 --warmup 1 --passes 2 > $O/${n}_next_masksidx.log 2>&1
done
for n in fetch write; do python scripts/pmc_summary.py $O/${n}_next_masksidx kernel | grep -v "^kernel," | sed "s/^/next_masksidx,/"; done > $O/pmc_traffic.csv
python scripts/make_traffic_json.py $O merge | tee $O/traffic.txt
cp profiles/hbm_traffic.json $O/hbm_traffic.json
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
( time python bench.py --steps 20 --warmup 3 --out $O/bench_default.jsonl ) > $O/bench_default.log 2> $O/bench_default.err
python bench.py --gpus 2 --backend gloo --single-device --no-cpu-baseline --steps 5 --warmup 2 2>$O/bench_gloo2.err | tail -1 > $O/bench_gloo2.json
python - <<'PY'
import json
for f in ("bench_default.jsonl", "bench_gloo2.json"):
    for l in open("gpurun_out/recheck6/" + f):
        if not l.strip():
            continue
        d = json.loads(l)
        print(d["config"]["workload"][:50], d["dtype"], "| %.1f G/s  %.2f ms/pass  kernel %.2f ms  frac %.3f  traffic %s" % (
            d["value"] / 1e9, d["config"]["ms_per_pass"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            None if d["roofline"]["traffic"] is None else round(d["roofline"]["traffic"] / 1e9, 2)), len(l))
print(d.get("cpu_baseline")); print(d["config"])
PY
tail -n 4 $O/bench_default.err; tail -n 3 $O/bench_gloo2.err | cut -c1-200
