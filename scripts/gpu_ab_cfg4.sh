#!/bin/bash
# A/B of library builds on the cfg4 bench lines at the full volume, interleaved (development aid):
#   bash scripts/gpu_ab_cfg4.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for wl in cfg4 "cfg4 --dtype float32" cfg4:planes64; do
    for L in "$@"; do
      ECHOPYPE_AMD_LIB=$L python bench.py --no-cpu-baseline --workload $wl --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], '$wl', d['dtype'], round(d['config']['ms_per_pass'],2), 'ms/pass', round(d['roofline']['kernel_ms'],2), 'ms kernel', round(d['roofline']['frac'],3))"
    done
  done
done
