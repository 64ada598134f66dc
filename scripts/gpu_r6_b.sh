#!/bin/bash
# round 6: full GPU suite + the depth bench lines with a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; rm -rf $O; mkdir -p $O
python scripts/perf_depth_fused.py 100000 2>&1 | grep -v "amdgpu.ids\|tau_eff" | tee $O/perf_100k.txt
for w in next:depth next:depthw cfg2 api cfg3 cfg3:ss2000 cfg3:f32 api:chain; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 >> $O/bench.jsonl 2>> $O/bench_err.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r6b/bench.jsonl"):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print(d["config"]["workload"][:60], "| Gs/s %.1f" % (d["value"]/1e9), "ms/pass %.3f" % d["config"]["ms_per_pass"], "kernel_ms %.3f" % d["roofline"].get("kernel_ms", -1), "frac %.3f" % d["roofline"]["frac"])
PY
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o k --output-format csv -- python bench.py --workload next:depth --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2> $O/kt_err.txt
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r6b/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
fused = [i for i, r in enumerate(rows) if "fused_sv_mvbs_kernel" in r[2]]
i0, i1 = fused[-4], fused[-2]
t0 = rows[i0][0]
out = open("gpurun_out/r6b/timeline.txt", "w")
prev = None
for s, e, n in rows[i0:i1 + 1]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap before {gap:7.1f}  {n[:70]}", file=out)
    prev = e
PY
cat $O/timeline.txt
find $O -name "*.csv" -size +1M -delete
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1
tail -5 $O/tests_all.txt
