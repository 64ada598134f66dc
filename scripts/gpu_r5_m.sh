#!/bin/bash
# cfg5 headline with the tiles dealt to K streams, results read L tiles late
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O; : > $O/tile_streams.txt
for cfg in "2 1" "2 2" "2 3" "3 2" "2 2" "1 1"; do
  set -- $cfg
  echo "== --tile-streams $1 --read-lag $2" >> $O/tile_streams.txt
  timeout 600 python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 --tile-streams $1 --read-lag $2 2>$O/err.txt | tail -1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value']/1e9, d['config']['ms_per_pass'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('host_ms_per_call'))" >> $O/tile_streams.txt
  grep -v amdgpu.ids $O/err.txt | tail -2 >> $O/tile_streams.txt
done
cat $O/tile_streams.txt
