#!/bin/bash
# cfg5 headline: tile size x streams x read lag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O; : > $O/tile_size.txt
for cfg in "250000 2 1" "125000 2 1" "125000 2 2" "125000 2 3" "125000 2 4" "125000 4 4" "125000 1 1" "250000 2 1" "125000 2 3"; do
  set -- $cfg
  echo "== --tile-pings $1 --tile-streams $2 --read-lag $3" >> $O/tile_size.txt
  timeout 600 python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 --tile-pings $1 --tile-streams $2 --read-lag $3 2>$O/err.txt | tail -1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value']/1e9, d['config']['ms_per_pass'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('host_ms_per_call'))" >> $O/tile_size.txt
  grep -v amdgpu.ids $O/err.txt | tail -2 >> $O/tile_size.txt
done
cat $O/tile_size.txt
