cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/masks4; rm -rf $O; mkdir -p $O
python scripts/perf_masks.py > $O/masks_probe.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python scripts/perf_masks.py > /dev/null 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -delete
grep -v amdgpu $O/masks_probe.txt | head -20
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/masks4/kernel_stats.csv")))
for r in rows[:28]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:9.3f} ms  {r['Percentage']:>6s}%")
PY
