#!/bin/bash
# After scripts/round5_final.sh came back through gpurun: copy what is judged from gpurun_out/final5/ into profiles/
F=gpurun_out/final5
[ -s $F/bench_default.jsonl ] && cp $F/bench_default.jsonl profiles/r05_bench_default.jsonl
[ -s $F/bench_gloo2.json ] && cp $F/bench_gloo2.json profiles/r05_bench_gloo2_single_gpu.json
[ -s $F/bench_sharded_at_1.json ] && cp $F/bench_sharded_at_1.json profiles/r05_bench_sharded_at_1.json
[ -s $F/kernel_stats.csv ] && cp $F/kernel_stats.csv profiles/r05_bench_kernel_stats.csv
[ -s $F/bench_under_rocprof.jsonl ] && cp $F/bench_under_rocprof.jsonl profiles/r05_bench_under_rocprof.jsonl
[ -s $F/pmc_traffic.csv ] && grep -E "epa_|sv_|block_reduce|power_coef|noise|mvbs|edge_|depth_rows|range_|impulse|attenuated|pool_|value_|row_|rows_|mask|nasc|minmax|step_|box_" $F/pmc_traffic.csv > profiles/r05_pmc_traffic.csv
[ -s $F/pmc_hot.csv ] && cp $F/pmc_hot.csv profiles/r05_pmc_hot.csv
[ -s $F/pmc_hot.txt ] && (cat $F/pmc_hot.txt; echo "(pmc_hot.py volumes: chain / fused 4 x 100 000 x 2000 = 0.8 G samples per launch; FFT 2 x 5000 x 8192 = 81.92 M output samples per launch)") > profiles/r05_pmc_hot.txt
[ -s $F/tests_gpu.txt ] && cp $F/tests_gpu.txt profiles/r05_tests_gpu.txt
[ -s $F/hbm_traffic.json ] && cp $F/hbm_traffic.json profiles/hbm_traffic.json   # carries the hash of the kernel sources it was measured on
[ -s gpurun_out/kernel_coverage.txt ] && cp gpurun_out/kernel_coverage.txt profiles/r05_kernel_coverage.txt
python scripts/kernel_resources.py > /dev/null 2>&1
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from bench import csrc_hash
t = json.load(open("profiles/hbm_traffic.json"))
print("csrc now", csrc_hash(), "| traffic measured at", sorted({v["csrc_sha16"] for v in t.values()}), "|", len(t), "keys")
PY
