#!/bin/bash
# round 6: the checked stream set of echopype_amd.pipeline -- its tests, then the default bench twice (does the headline hold its
# two-stream figure at the end of a long process?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6pairs; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_streams.py tests/test_bench_contract.py tests/test_gpu_pipelined_api.py -q -m gpu -x > $O/tests.txt 2>&1; tail -n 4 $O/tests.txt
for i in 1 2; do
  ( time python bench.py --steps 20 --warmup 3 --out $O/bench_default_$i.jsonl ) > $O/bench_$i.log 2> $O/bench_$i.err
  tail -n 2 $O/bench_default_$i.jsonl | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l); c = d['config']; r = d['roofline']
    print('%.1f G/s  ms/pass %.2f  each %.2f  frac %.3f  chars %d  streams %s' % (d['value'] / 1e9, c['ms_per_pass'], r.get('kernel_ms_each', 0), r['frac'], len(l), c.get('tile_streams')))"
done
