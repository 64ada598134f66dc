#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
timeout 900 python bench.py --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
