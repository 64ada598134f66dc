#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ek80api; mkdir -p $O
python -m pytest tests -q -m gpu -k "ek80 or EK80 or complex or multi_filter or xarray or smooth or nasc or fullsize_chain" 2>&1 | tail -12 > $O/tests.txt; tail -6 $O/tests.txt
python scripts/perf_cfg4.py 2 200000 8192 4 > $O/cfg4_full.txt 2>&1; cat $O/cfg4_full.txt | tail -12
