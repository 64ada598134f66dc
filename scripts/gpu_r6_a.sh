#!/bin/bash
# round 6: the fused depth route (A/B at ops level, tests, bench lines)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; rm -rf $O; mkdir -p $O
python scripts/perf_depth_fused.py 100000 2>&1 | grep -v "amdgpu.ids\|tau_eff" | tee $O/perf_100k.txt
python scripts/perf_depth_fused.py 500000 2>&1 | grep -v "amdgpu.ids\|tau_eff" | tee $O/perf_500k.txt
timeout 1500 python -m pytest tests/test_gpu_depth_fused.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_kernels.py tests/test_gpu_pipelined_api.py tests/test_gpu_streams.py -x -q -m gpu > $O/tests.txt 2>&1
tail -8 $O/tests.txt
for w in next:depth next:depthw cfg2 api; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 >> $O/bench.jsonl 2>> $O/bench_err.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r6a/bench.jsonl"):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print(d["config"]["workload"][:60], "| Gs/s %.1f" % (d["value"]/1e9), "ms/pass %.3f" % d["config"]["ms_per_pass"], "kernel_ms %.3f" % d["roofline"].get("kernel_ms", -1), "frac %.3f" % d["roofline"]["frac"])
PY
