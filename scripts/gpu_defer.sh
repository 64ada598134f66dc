#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/defer
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/defer/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/defer/pytest.log
timeout 600 python bench.py --workload api --steps 10 --warmup 2 > gpurun_out/defer/api.log 2>&1
echo "bench rc=$?"; tail -c 1800 gpurun_out/defer/api.log
timeout 600 python bench.py --workload cfg2:f32 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
