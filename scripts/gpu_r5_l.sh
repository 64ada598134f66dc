#!/bin/bash
# the fused kernel as several launches side by side (EPA_FUSED_STREAMS) / interleaved runs per XCD (EPA_XCD_WAYS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O; : > $O/ways.txt; : > $O/err.txt
EPA_FUSED_STREAMS=4 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/tests.txt
for cfg in "1 1" "2 1" "4 1" "8 1" "1 1" "4 1" "4 2" "2 2"; do
  set -- $cfg
  echo "== EPA_FUSED_STREAMS=$1 EPA_XCD_WAYS=$2" >> $O/ways.txt
  for wl in cfg2 cfg2:f32 cfg2:int16; do
  EPA_FUSED_STREAMS=$1 EPA_XCD_WAYS=$2 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 2>>$O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:24], d['dtype'], d['config']['ms_per_pass'], d['roofline']['kernel_ms'], d['roofline']['frac'])" >> $O/ways.txt
  done
done
cat $O/ways.txt; tail -3 $O/err.txt
