#!/bin/bash
# Round-end GPU call: whole -m gpu suite, default bench, 2-rank gloo dry run, rocprofv3 kernel stats of the default bench,
# PMC HBM-traffic passes, PMC instruction counters of the hot kernels, API and mask probes.  Results under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/tests.txt; tail -3 $O/tests.txt
for wl in cfg2 cfg3 cfg4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    n=fetch; [ $c = "FETCH_SIZE" ] || n=write
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/${n}_$wl -o p --output-format csv -- python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > $O/${n}_$wl.log 2>&1
  done
done
python scripts/make_traffic_json.py $O | tee $O/traffic.txt
cp profiles/hbm_traffic.json $O/hbm_traffic.json
# (the bench lines come after the traffic measurement: they carry it, stamped with the hash of the sources)
python bench.py > $O/bench_default.jsonl 2> $O/bench_default.err
python bench.py --gpus 2 --backend gloo --single-device --pings-total 400000 > $O/bench_gloo2.json 2> $O/bench_gloo2.err
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.jsonl 2> $O/bench_under_rocprof.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -delete
python scripts/perf_api_resident.py > $O/api_resident.txt 2>&1
python scripts/perf_api_two_calls.py 2>&1 | grep -v "^$" | cut -c1-160 > $O/api_two_calls.txt
python scripts/perf_masks.py > $O/masks_probe.txt 2>&1
(python scripts/perf_cw.py 2>&1 | grep -v amdgpu.ids; echo "== scripts/probes/hbm_ek80_probe.hip"; [ -x scripts/probes/bin/hbm_ek80_probe ] && scripts/probes/bin/hbm_ek80_probe) > $O/cw_probe.txt 2>&1
python scripts/pmc_summary.py $O/fetch_cfg2 > $O/pmc_traffic_a.csv 2>/dev/null
for wl in cfg2 cfg3 cfg4; do for n in fetch write; do python scripts/pmc_summary.py $O/${n}_$wl kernel | grep -v "^kernel," | sed "s/^/$wl,/" ; done; done > $O/pmc_traffic.csv
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
bash scripts/gpu_pmc_hot.sh all > $O/pmc_hot.txt 2>&1; cp gpurun_out/pmc_hot/summary.csv $O/pmc_hot.csv
python - <<'PY'
import json
for l in open("gpurun_out/final/bench_default.jsonl"):
    d = json.loads(l); c = d["config"]
    print(d["config"]["workload"][:60], d["dtype"], "| %.1f G/s  %.2f ms/step  kernel %.2f ms  frac %.3f  cpu %.3g" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value", 0)), {k: round(v, 2) for k, v in c.items() if k.endswith("ms_per_step")})
PY
tail -3 $O/pmc_hot.txt; cat $O/traffic.txt | cut -c1-200
