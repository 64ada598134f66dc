#!/bin/bash
# round 5: the row-major fused kernel against the chunk-walking one (EPA_FUSED_ROWS=0), then the GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do
  for R in 1 0; do
    echo "== EPA_FUSED_ROWS=$R cfg5 tile"; EPA_FUSED_ROWS=$R python scripts/perf_fused.py 4 250000 4096 2>&1 | grep -v amdgpu.ids
    echo "== EPA_FUSED_ROWS=$R cfg2"; EPA_FUSED_ROWS=$R python scripts/perf_fused.py 4 500000 2000 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r5b_fused_ab.txt 2>&1
cat gpurun_out/r5b_fused_ab.txt
python -m pytest tests -m gpu -q --maxfail=40 -x -k "not multi_rank and not sharded_sonars" > gpurun_out/r5b_tests.txt 2>&1; echo "tests rc $?"
tail -25 gpurun_out/r5b_tests.txt
python -m pytest tests/test_gpu_multi_rank.py tests/test_gpu_sharded_sonars.py -m gpu -q --maxfail=40 > gpurun_out/r5b_tests_shard.txt 2>&1; echo "shard tests rc $?"
tail -40 gpurun_out/r5b_tests_shard.txt
