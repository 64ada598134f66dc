#!/bin/bash
# the fused kernel with the channels side by side (EPA_FUSED_CHANPAR=<ways>)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O; : > $O/chanpar.txt; : > $O/err.txt
EPA_FUSED_CHANPAR=8 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/tests.txt
for w in 0 1 8 9 1 8 9; do
  echo "== EPA_FUSED_CHANPAR=$w" >> $O/chanpar.txt
  for wl in cfg2 cfg2:f32; do
  EPA_FUSED_CHANPAR=$w python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 2>>$O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:24], d['dtype'], d['config']['ms_per_pass'], d['roofline']['kernel_ms'], d['roofline']['frac'])" >> $O/chanpar.txt
  done
done
EPA_FUSED_CHANPAR=8 python scripts/perf_stagger.py 2>&1 | grep "stream" | head -3 >> $O/chanpar.txt
cat $O/chanpar.txt; tail -3 $O/err.txt
