#!/bin/bash
# round 6: when does the headline lose its two-stream gain inside one process?  The headline after different line sequences.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6seq; rm -rf $O; mkdir -p $O
show() { tail -n 1 | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l); c = d['config']; r = d['roofline']
    print('   -> %.1f G/s  ms/pass %.2f  each %.2f  frac %.3f' % (d['value'] / 1e9, c['ms_per_pass'], r.get('kernel_ms_each', 0), r['frac']))"; }
for seq in "cfg5" "cfg5:one,cfg5" "cfg4,cfg5" "next:masks2000,cfg5" "api:pcie,cfg5" "cfg3,cfg2,cfg5" "cfg5:one,cfg5"; do
  echo "== $seq"
  python bench.py --workload $seq --no-cpu-baseline --steps 10 --warmup 3 2> $O/err.txt | show
done
echo "== EPA_BENCH_KEEP=1 cfg5:one,cfg5 (a read result kept over the next launch, as before)"
EPA_BENCH_KEEP=1 python bench.py --workload cfg5:one,cfg5 --no-cpu-baseline --steps 10 --warmup 3 2> $O/err.txt | show
EPA_BENCH_KEEP=1 python bench.py --workload cfg3,cfg2,cfg5 --no-cpu-baseline --steps 10 --warmup 3 2> $O/err.txt | show
