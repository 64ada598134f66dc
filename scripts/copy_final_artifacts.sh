#!/bin/bash
# After scripts/final_round.sh came back through gpurun: copy what is judged from gpurun_out/final/ into profiles/
F=gpurun_out/final
cp $F/bench_default.jsonl profiles/r02_bench_default.jsonl
cp $F/bench_gloo2.json profiles/r02_bench_gloo2_single_gpu.json
cp $F/kernel_stats.csv profiles/r02_bench_kernel_stats.csv
cp $F/bench_under_rocprof.jsonl profiles/r02_bench_under_rocprof.jsonl
grep -E "epa_|sv_complex|block_reduce|power_coef|noise|mvbs" $F/pmc_traffic.csv > profiles/r02_pmc_traffic.csv
cp $F/pmc_hot.csv profiles/r02_pmc_hot.csv
(cat $F/pmc_hot.txt; echo "(pmc_hot.py volumes: chain / fused 4 x 100 000 x 2000 = 0.8 G samples per launch; FFT 2 x 5000 x 8192 = 81.92 M output samples per launch)") > profiles/r02_pmc_hot.txt
cp $F/tests.txt profiles/r02_tests_gpu.txt
cp $F/api_resident.txt profiles/r02_api_resident.txt
cp $F/api_two_calls.txt profiles/r02_api_two_calls.txt
cp $F/masks_probe.txt profiles/r02_masks_probe.txt
cp $F/cw_probe.txt profiles/r02_ek80_cw_probe.txt
cp $F/hbm_traffic.json profiles/hbm_traffic.json   # carries the hash of the kernel sources it was measured on
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from bench import csrc_hash
t = json.load(open("profiles/hbm_traffic.json"))
print("csrc now", csrc_hash(), "| traffic measured at", sorted({v["csrc_sha16"] for v in t.values()}))
PY
