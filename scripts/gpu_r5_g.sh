#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_nasc.py tests/test_gpu_kernels.py -m gpu -q --maxfail=10 > $O/tests.txt 2>&1; echo "tests rc $?"; tail -6 $O/tests.txt | cut -c1-200
python bench.py --workload next:nasc --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], '%.2f ms/pass frac %.3f' % (d['config']['ms_per_pass'], d['roofline']['frac']))"
