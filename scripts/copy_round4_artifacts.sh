#!/bin/bash
# After scripts/round4_final.sh came back through gpurun: copy what is judged from gpurun_out/final4/ into profiles/
F=gpurun_out/final4
[ -s $F/bench_default.jsonl ] && cp $F/bench_default.jsonl profiles/r04_bench_default.jsonl
[ -s $F/bench_gloo2.json ] && cp $F/bench_gloo2.json profiles/r04_bench_gloo2_single_gpu.json
[ -s $F/kernel_stats.csv ] && cp $F/kernel_stats.csv profiles/r04_bench_kernel_stats.csv
[ -s $F/bench_under_rocprof.jsonl ] && cp $F/bench_under_rocprof.jsonl profiles/r04_bench_under_rocprof.jsonl
[ -s $F/pmc_traffic.csv ] && grep -E "epa_|sv_complex|block_reduce|power_coef|noise|mvbs|edge_" $F/pmc_traffic.csv > profiles/r04_pmc_traffic.csv
[ -s $F/pmc_hot.csv ] && cp $F/pmc_hot.csv profiles/r04_pmc_hot.csv
[ -s $F/pmc_hot.txt ] && (cat $F/pmc_hot.txt; echo "(pmc_hot.py volumes: chain / fused 4 x 100 000 x 2000 = 0.8 G samples per launch; FFT 2 x 5000 x 8192 = 81.92 M output samples per launch)") > profiles/r04_pmc_hot.txt
[ -s $F/api_cfg5_probe.txt ] && grep -v amdgpu.ids $F/api_cfg5_probe.txt > profiles/r04_api_cfg5_probe.txt
[ -s $F/tests.txt ] && cp $F/tests.txt profiles/r04_tests_gpu.txt
[ -s $F/hbm_traffic.json ] && cp $F/hbm_traffic.json profiles/hbm_traffic.json   # carries the hash of the kernel sources it was measured on
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from bench import csrc_hash
t = json.load(open("profiles/hbm_traffic.json"))
print("csrc now", csrc_hash(), "| traffic measured at", sorted({v["csrc_sha16"] for v in t.values()}), "|", len(t), "keys")
PY
