#!/bin/bash
# cfg5 headline with the tiles dealt to K streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O; : > $O/tile_streams.txt
for k in 2 3 4 2 3 4 1; do
  echo "== --tile-streams $k" >> $O/tile_streams.txt
  timeout 600 python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 --tile-streams $k 2>$O/err_$k.txt | tail -1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value']/1e9, d['config']['ms_per_pass'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('host_ms_per_call'))" >> $O/tile_streams.txt
  tail -2 $O/err_$k.txt >> $O/tile_streams.txt
done
cat $O/tile_streams.txt
