#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/api; mkdir -p $O
python -m pytest tests -q -m gpu -k "api or fuzz or ek80 or complex or sharded or fullsize_chain" 2>&1 | tail -12 > $O/tests.txt; tail -6 $O/tests.txt
python scripts/perf_api_profile.py > $O/profile.txt 2>&1; grep -E "parameters|cumtime|echopype_amd|method" $O/profile.txt | head -45
bash scripts/gpu_pmc_hot.sh fft 2>&1 | tail -3
python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 f64', d['roofline']['kernel_ms'], d['roofline']['frac'])"
python bench.py --workload cfg4 --no-cpu-baseline --dtype float32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 f32', d['roofline']['kernel_ms'], d['roofline']['frac'])"
