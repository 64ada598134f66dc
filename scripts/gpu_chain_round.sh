#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/chain; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_chain.py tests/test_gpu_sharded.py -q -m gpu -k "chain or two_pass or clean or random_shapes or bb_windows or cfg5_range or sharded or ek60 or EK60" 2>&1 | tail -30 > $O/tests.txt; tail -8 $O/tests.txt
python scripts/perf_chain.py > $O/probe.txt 2>&1; grep -E "pass|two passes" $O/probe.txt | head -8
bash scripts/gpu_pmc_hot.sh chain 2>&1 | tail -4
python scripts/perf_api_resident.py 2>&1 | grep -E "compute_Sv|compute_Sv_MVBS" | head -4
