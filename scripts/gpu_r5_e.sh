#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; rm -rf $O; mkdir -p $O
for i in 1 2; do for k in 1 0; do EPA_K1_PIECES=$k python scripts/perf_k1_pieces.py 2>&1 | grep -v amdgpu.ids; done; done > $O/k1_pieces_ab.txt; cat $O/k1_pieces_ab.txt
python -m pytest tests -m gpu -q --maxfail=40  > $O/tests.txt 2>&1; echo "tests rc $?"; tail -30 $O/tests.txt | cut -c1-220
