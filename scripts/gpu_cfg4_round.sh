#!/bin/bash
# EK80 tests + the cfg4 bench lines (fp64 and fp32 output) -- development aid
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu -k "complex or ek80 or fft or bb" 2>&1 | tail -3
for dt in float64 float32; do
python bench.py --workload cfg4 --dtype $dt --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['dtype'], '| %.2f ms kernel %.2f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
"
done
