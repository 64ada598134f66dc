"""Development aid: from a rocprofv3 kernel trace of scripts/perf_api_cfg5.py -- how long the fused kernel runs and how
long the compute queue spends between two of its launches (small kernels + dispatch gaps), over all launches.
    python scripts/trace_api_gaps.py <kernel_trace.csv>"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if "fused_sv_mvbs_kernel" in r["Kernel_Name"]]
dur = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big]) / 1e3
gap = np.array([int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(big[:-1], big[1:])]) / 1e3
gap = gap[gap < 3000]   # (pauses between the script's sections)
q = lambda a: " ".join(f"{np.percentile(a, p):8.1f}" for p in (5, 25, 50, 75, 95))
print(f"{len(big)} launches of the fused kernel; percentiles 5 / 25 / 50 / 75 / 95 (us)")
print("kernel duration     ", q(dur))
print("between two launches", q(gap))
small = {}
for r in rows:
    if "fused_sv_mvbs_kernel" in r["Kernel_Name"]:
        continue
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    small.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(small.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"   {n:44s} x{len(v) / max(1, len(big)):5.1f} per launch, {np.mean(v):7.1f} us each")
