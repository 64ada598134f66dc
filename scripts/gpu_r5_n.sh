#!/bin/bash
# kernel timeline of the headline loop with two tile streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o k --output-format csv -- python bench.py --only-headline --no-cpu-baseline --steps 3 --warmup 1 --tile-streams 2 --read-lag ${LAG:-1} > $O/line.json 2> $O/err.txt
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r5n/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ks.sort()
fused = [k for k in ks if "fused_sv_mvbs_kernel" in k[2]]
t0 = fused[-40][0]
out = open("gpurun_out/r5n/timeline.txt", "w")
prev_end = None
for s, e, n, q in fused[-40:]:
    print(f"start {(s - t0) / 1e6:9.3f} ms  end {(e - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e6:7.3f}  queue {q}", file=out)
# GPU idle time and single-kernel time over the last 32 launches
ev = sorted([(s, 1) for s, e, n, q in fused[-32:]] + [(e, -1) for s, e, n, q in fused[-32:]])
lvl, last, acc = 0, ev[0][0], {0: 0, 1: 0, 2: 0, 3: 0}
for tt, d in ev:
    acc[min(lvl, 3)] += tt - last
    last = tt
    lvl += d
tot = ev[-1][0] - ev[0][0]
print(f"last 32 launches: {tot / 1e6 / 32:.3f} ms per launch; no fused kernel running {acc[0] / tot:.3f}, one {acc[1] / tot:.3f}, two {acc[2] / tot:.3f}", file=out)
PY
cat $O/timeline.txt | tail -24; find $O -name "*.csv" -size +1M -delete
