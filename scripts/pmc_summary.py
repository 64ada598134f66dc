"""Summarise a rocprofv3 counter-collection CSV: mean counter value per (kernel, counter) for kernels whose name
contains one of the given substrings (development aid; output goes to profiles/).
    python scripts/pmc_summary.py <dir or csv> [substring ...] > profiles/rNN_pmc_x.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
subs = sys.argv[2:] or ["epa", "kernel"]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(list)
meta = {}
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if not any(s in k for s in subs) or k.startswith("void at::"):
                continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-110:]
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
print("kernel,Counter_Name,mean_value,dispatches,vgpr,lds,scratch")
for (k, c), v in sorted(acc.items()):
    print(f'"{k}",{c},{sum(v) / len(v)},{len(v)},{meta[k][0]},{meta[k][1]},{meta[k][2]}')
