#!/bin/bash
# Round profile: rocprofv3 kernel stats of the default bench + PMC HBM traffic of the three workloads (separate passes).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.jsonl 2> $O/bench_under_rocprof.err
for wl in cfg2 cfg3 cfg4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | cut -c1-5 | tr 'A-Z' 'a-z'); [ $n = "fetch" ] || n=write
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/${n}_$wl -o p --output-format csv -- python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > $O/${n}_$wl.log 2>&1
  done
done
python scripts/make_traffic_json.py $O | tee $O/traffic.txt
cp profiles/hbm_traffic.json $O/hbm_traffic.json
python scripts/pmc_summary.py $O/fetch_cfg2 $O/write_cfg2 $O/fetch_cfg3 $O/write_cfg3 $O/fetch_cfg4 $O/write_cfg4 > /dev/null 2>&1
ls $O/kt
