#!/bin/bash
# A/B several builds of the library with bench.py, interleaved rounds (development aid)
ROUNDS=${ROUNDS:-2}
for i in $(seq $ROUNDS); do
  for L in "$@"; do
    ECHOPYPE_AMD_LIB=$L python bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], round(d['value']/1e9,1), 'Gs/s', round(d['roofline']['kernel_ms'],3), 'ms kernel')"
  done
done
