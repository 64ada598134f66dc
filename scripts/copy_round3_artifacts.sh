#!/bin/bash
# After scripts/round3_final.sh came back through gpurun: copy what is judged from gpurun_out/final3/ into profiles/
F=gpurun_out/final3
cp $F/bench_default.jsonl profiles/r03_bench_default.jsonl
cp $F/bench_gloo2.json profiles/r03_bench_gloo2_single_gpu.json
cp $F/kernel_stats.csv profiles/r03_bench_kernel_stats.csv
cp $F/bench_under_rocprof.jsonl profiles/r03_bench_under_rocprof.jsonl
grep -E "epa_|sv_complex|block_reduce|power_coef|noise|mvbs|edge_" $F/pmc_traffic.csv > profiles/r03_pmc_traffic.csv
cp $F/pmc_hot.csv profiles/r03_pmc_hot.csv
(cat $F/pmc_hot.txt; echo "(pmc_hot.py volumes: chain / fused 4 x 100 000 x 2000 = 0.8 G samples per launch; FFT 2 x 5000 x 8192 = 81.92 M output samples per launch)") > profiles/r03_pmc_hot.txt
cp $F/tests.txt profiles/r03_tests_gpu.txt
[ -f $F/pmc_fft_stalls.csv ] && cp $F/pmc_fft_stalls.csv profiles/r03_pmc_fft_stalls.csv
cp $F/hbm_traffic.json profiles/hbm_traffic.json   # carries the hash of the kernel sources it was measured on
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from bench import csrc_hash
t = json.load(open("profiles/hbm_traffic.json"))
print("csrc now", csrc_hash(), "| traffic measured at", sorted({v["csrc_sha16"] for v in t.values()}))
PY
