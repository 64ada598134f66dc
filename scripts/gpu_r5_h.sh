#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -m gpu -q --maxfail=10 > $O/tests.txt 2>&1; echo "tests rc $?"; tail -5 $O/tests.txt | cut -c1-200
for i in 1 2; do for k in 1 0; do EPA_K1_PIECES=$k python scripts/perf_k1_pieces.py 2>&1 | grep -v amdgpu.ids | grep -E "stats|Sv  "; done; done > $O/k1_stats_ab.txt; cat $O/k1_stats_ab.txt
