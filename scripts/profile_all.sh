#!/bin/bash
# rocprofv3 kernel-trace summaries of every kernel of the path (development / profiles/ aid).
# usage (on the GPU box): scripts/profile_all.sh gpurun_out/prof_all
OUT=${1:-gpurun_out/prof_all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/ek60" -o k --output-format csv -- python scripts/perf_probe.py 4 100000 2000 > "$OUT/ek60_stdout.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/ek80" -o k --output-format csv -- python scripts/perf_ek80.py > "$OUT/ek80_stdout.txt" 2>&1
grep -E "TB/s" "$OUT/ek60_stdout.txt"; grep -E "^BB|^CW" "$OUT/ek80_stdout.txt"
