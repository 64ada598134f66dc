#!/bin/bash
# A/B of an environment knob on the same box: scripts/gpu_ab_env.sh VAR "bench args" [rounds]  (development aid)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VAR=$1; ARGS=$2; ROUNDS=${3:-3}
for i in $(seq $ROUNDS); do
  for v in 0 1; do
    env $VAR=$v python bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']/1e9,1), 'Gs/s', round(d['roofline']['kernel_ms'],3), 'ms kernel, frac', round(d['roofline']['frac'],3))"
  done
done
