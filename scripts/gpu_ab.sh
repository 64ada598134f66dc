#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BENCH_ARGS="--only-headline --steps 100 --warmup 10"
ROUNDS=3 bash scripts/ab_bench.sh echopype_amd/lib/libechopype_amd.so "$@"
