#!/bin/bash
# mask tests on the default library, then the value-window pooling timings of the named variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_masks.py tests/test_gpu_masks_api.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?"; tail -5 $O/tests.txt
bash scripts/gpu_r5_i.sh "$@"
echo "== base, EPA_POOL_LEAN=0" >> gpurun_out/r5i/pool_value_ab.txt
EPA_POOL_LEAN=0 timeout 600 python scripts/perf_pool_value.py >> gpurun_out/r5i/pool_value_ab.txt 2>&1
tail -8 gpurun_out/r5i/pool_value_ab.txt
