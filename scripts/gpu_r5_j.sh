#!/bin/bash
# mask tests on the default library (or the library named by TEST_LIB), then the value-window pooling timings of the named variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
if [ -n "$TEST_LIB" ]; then export ECHOPYPE_AMD_LIB=$PWD/echopype_amd/lib/libechopype_amd_$TEST_LIB.so; fi
timeout 1500 python -m pytest tests/test_gpu_masks.py tests/test_gpu_masks_api.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?"; tail -5 $O/tests.txt
unset ECHOPYPE_AMD_LIB
bash scripts/gpu_r5_i.sh "$@"
