#!/bin/bash
# After scripts/round6_final.sh came back through gpurun: copy what is judged from gpurun_out/final6/ into profiles/
F=gpurun_out/final6
[ -s $F/bench_default.jsonl ] && cp $F/bench_default.jsonl profiles/r06_bench_default.jsonl
[ -s $F/bench_headline.json ] && cp $F/bench_headline.json profiles/r06_bench_headline_alone.json
[ -s $F/bench_gloo2.json ] && cp $F/bench_gloo2.json profiles/r06_bench_gloo2_single_gpu.json
[ -s $F/bench_sharded_at_1.json ] && cp $F/bench_sharded_at_1.json profiles/r06_bench_sharded_at_1.json
[ -s $F/kernel_stats.csv ] && cp $F/kernel_stats.csv profiles/r06_bench_kernel_stats.csv
[ -s $F/bench_under_rocprof.jsonl ] && cp $F/bench_under_rocprof.jsonl profiles/r06_bench_under_rocprof.jsonl
[ -s $F/pmc_traffic.csv ] && grep -E "epa_|sv_|block_reduce|power_coef|noise|mvbs|edge_|depth_rows|range_|impulse|attenuated|pool_|value_|row_|rows_|mask|nasc|minmax|step_|box_|run_|piece" $F/pmc_traffic.csv > profiles/r06_pmc_traffic.csv
[ -s $F/traffic.txt ] && cp $F/traffic.txt profiles/r06_traffic_ratios.txt
[ -s $F/tests_gpu.txt ] && cp $F/tests_gpu.txt profiles/r06_tests_gpu.txt
[ -s $F/hbm_traffic.json ] && cp $F/hbm_traffic.json profiles/hbm_traffic.json   # carries the hash of the kernel sources it was measured on
[ -s gpurun_out/kernel_coverage.txt ] && cp gpurun_out/kernel_coverage.txt profiles/r06_kernel_coverage.txt
python scripts/kernel_resources.py > /dev/null 2>&1
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from bench import csrc_hash
t = json.load(open("profiles/hbm_traffic.json"))
print("csrc now", csrc_hash(), "| traffic measured at", sorted({v["csrc_sha16"] for v in t.values()}), "|", len(t), "keys")
PY
