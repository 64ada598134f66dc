"""Scan the compiler's assembly of every kernel of the library for global loads that are separated from the previous
load by an ``s_waitcnt vmcnt(0)`` and few vector instructions -- requests that go out one round trip at a time
(typically: an unrolled loop whose loads sit behind per-element branches).  No GPU needed.
    python scripts/scan_serialized_loads.py [max VALU between the loads, default 40] [min hits, default 4]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from echopype_amd import build  # noqa: E402

max_valu = int(sys.argv[1]) if len(sys.argv) > 1 else 40
min_hits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out = tempfile.mkdtemp(prefix="epa_asm_")
procs = []
for src in build.SOURCES:
    cmd = [build._hipcc(), *[f for f in build.FLAGS if f not in ("-Wall", "-fPIC")], "--offload-device-only", "-S",
           os.path.join(build.CSRC, src), "-o", os.path.join(out, src.replace(".hip", ".s"))]
    procs.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
for path in sorted(glob.glob(os.path.join(out, "*.s"))):
    lines = open(path).read().split("\n")
    for start in [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]:
        end = next((i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        seen, valu, waited, hits, loads = False, 0, False, 0, 0
        for l in lines[start:end]:
            s = l.strip()
            op = s.split(" ")[0]
            if op.startswith("v_"):
                valu += 1
            if op == "s_waitcnt" and "vmcnt(0)" in s:
                waited = True
            if op.startswith(("global_load", "buffer_load")):
                loads += 1
                if seen and waited and valu <= max_valu:
                    hits += 1
                seen, valu, waited = True, 0, False
        if hits >= min_hits:
            name = subprocess.run(["c++filt", lines[start].split(":")[0]], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(path):20s} {hits:3d} of {loads:3d} loads  {name[:120]}")
