#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/masks_r2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_masks.py tests/test_gpu_masks_api.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_next_rows.py -q -m gpu -k "mask or atten" 2>&1 | tail -25 > $O/tests.txt; tail -12 $O/tests.txt | cut -c1-200
timeout 300 python scripts/perf_masks.py > $O/probe.txt 2>&1; grep -E "atten|impulse|transient|apply" $O/probe.txt | head
