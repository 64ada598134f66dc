#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; rm -rf $O; mkdir -p $O
for i in 1 2; do for k in 1 0; do EPA_K1_PIECES=$k python scripts/perf_k1_pieces.py 2>&1 | grep -v amdgpu.ids | grep -E "stats|Sv  "; done; done > $O/k1_stats_ab.txt; cat $O/k1_stats_ab.txt
for i in 1 2; do
  for L in libechopype_amd.so libechopype_amd_r4fused.so; do
    echo "== $L"; ECHOPYPE_AMD_LIB=echopype_amd/lib/$L python scripts/perf_fused.py 4 250000 4096 2>&1 | grep -v amdgpu.ids | grep float32
    ECHOPYPE_AMD_LIB=echopype_amd/lib/$L python scripts/perf_fused.py 4 500000 2000 2>&1 | grep -v amdgpu.ids | grep float32
  done
done > $O/fused_f32_quad_ab.txt 2>&1; cat $O/fused_f32_quad_ab.txt
python -m pytest tests -m gpu -q --maxfail=40 -k "not multi_rank and not sharded and not masks and not nasc" > $O/tests.txt 2>&1; echo "tests rc $?"; tail -12 $O/tests.txt | cut -c1-220
