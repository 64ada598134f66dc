#!/bin/bash
# round 5, first GPU call: the whole GPU suite on the new tree + A/B of the fused kernel (lean path) against round 4's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/r5a_tests.txt 2>&1; echo "tests rc $?"
tail -15 gpurun_out/r5a_tests.txt
for i in 1 2; do
  for L in libechopype_amd.so libechopype_amd_r4fused.so; do
    echo "== $L"; ECHOPYPE_AMD_LIB=echopype_amd/lib/$L python scripts/perf_fused.py 4 250000 4096 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r5a_fused_ab.txt 2>&1
cat gpurun_out/r5a_fused_ab.txt
