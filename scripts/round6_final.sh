#!/bin/bash
# Round-6 final GPU call: PMC HBM-traffic passes for EVERY default bench line (separate --pmc passes, --kernel-trace
# only), the default bench (its lines carry the traffic just measured), the sharded route on a one-rank RCCL group and the
# 2-rank gloo dry run at the real tile size, rocprofv3 kernel stats of the default bench, the GPU test suite.
# Results under gpurun_out/final6/ (scripts/copy_round6_artifacts.sh takes them to profiles/).
#   SKIP_TESTS=1 leaves the test suite out, SKIP_PMC=1 the counter passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final6; rm -rf $O; mkdir -p $O
LINES="cfg2 cfg2:f32 cfg2:int16 cfg2:int16f32 cfg2:bins cfg2:int16bins cfg2:sv cfg2:sv32 cfg3 cfg3:f32 cfg3:ss2000 cfg4 cfg4:f32 cfg4:planes64 cfg5 api api:chain next:depth next:depthw next:masks next:masks2000 next:masksidx next:nasc"
if [ -z "$SKIP_PMC" ]; then
for wl in $LINES; do
  tag=$(echo $wl | tr ':' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    n=fetch; [ $c = "FETCH_SIZE" ] || n=write
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/${n}_$tag -o p --output-format csv -- python bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 1 --passes 2 > $O/${n}_$tag.log 2>&1
  done
done
for wl in $LINES; do tag=$(echo $wl | tr ':' '_'); for n in fetch write; do python scripts/pmc_summary.py $O/${n}_$tag kernel | grep -v "^kernel," | sed "s/^/$tag,/" ; done; done > $O/pmc_traffic.csv
python scripts/make_traffic_json.py $O | tee $O/traffic.txt
cp profiles/hbm_traffic.json $O/hbm_traffic.json
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
fi
# (the bench lines come after the traffic measurement: they carry it, stamped with the hash of the sources)
python bench.py --steps 20 --warmup 3 --out $O/bench_default.jsonl > $O/bench_default.log 2> $O/bench_default.err
python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 2>$O/bench_headline.err | tail -1 > $O/bench_headline.json
python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 --sharded-at-1 2>$O/bench_sharded_at_1.err | tail -1 > $O/bench_sharded_at_1.json
python bench.py --gpus 2 --backend gloo --single-device --no-cpu-baseline --steps 5 --warmup 2 2>$O/bench_gloo2.err | tail -1 > $O/bench_gloo2.json
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_under_rocprof.jsonl 2> $O/bench_under_rocprof.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -delete
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -q > $O/tests_gpu.txt 2>&1; tail -n 3 $O/tests_gpu.txt; fi
python - <<'PY'
import json
for f in ("bench_default.jsonl", "bench_headline.json", "bench_sharded_at_1.json"):
    for l in open("gpurun_out/final6/" + f):
        if not l.strip():
            continue
        d = json.loads(l)
        print(d["config"]["workload"][:50], d["dtype"], "| %.1f G/s  %.2f ms/pass  kernel %.2f ms  frac %.3f  traffic %s" % (
            d["value"] / 1e9, d["config"]["ms_per_pass"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            None if d["roofline"]["traffic"] is None else round(d["roofline"]["traffic"] / 1e9, 2)), len(l))
PY
[ -s $O/traffic.txt ] && cut -c1-160 $O/traffic.txt
