"""Registers, scratch, LDS and occupancy of every kernel of the library as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; no GPU needed) -> profiles/r06_kernel_resources.txt.  rocprofv3's Arch_VGPR column
counts register PAIRS on wave64 (half of "VGPRs" here)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from echopype_amd import build  # noqa: E402

rows = []
for src in build.SOURCES:
    cmd = [build._hipcc(), *[f for f in build.FLAGS if f != "-Wall"], "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(build.CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, v = None, {}
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur, v = m.group(1), {}
        for key in ("SGPRs:", "VGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "LDS Size [bytes/block]:"):
            if key in line and cur:
                v[key] = line.split(key)[1].split("[")[0].strip()
        if "LDS Size" in line and cur:
            # (strip the namespace BEFORE cutting the argument list off: "(anonymous namespace)::k<..>(Args)" starts
            #  with a parenthesis -- round 3's table lost 201 of its 268 names to that)
            full = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()
            full = full.replace("(anonymous namespace)::", "").replace("void ", "")
            depth, cut = 0, len(full)
            for i, ch in enumerate(full):  # the argument list opens at the first "(" outside template brackets
                if ch == "<":
                    depth += 1
                elif ch == ">":
                    depth -= 1
                elif ch == "(" and depth == 0:
                    cut = i
                    break
            name = full[:cut].strip() or cur
            rows.append((src, name, v))
            cur = None
out = [__doc__.strip(), "", f"{'kernel':92s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'waves/SIMD':>10s} {'static LDS':>10s}"]
for src, name, v in sorted(rows, key=lambda r: (r[0], r[1])):
    out.append(f"{name[:92]:92s} {v.get('VGPRs:', '?'):>5s} {v.get('SGPRs:', '?'):>5s} {v.get('ScratchSize [bytes/lane]:', '?'):>8s} "
               f"{v.get('Occupancy [waves/SIMD]:', '?'):>10s} {v.get('LDS Size [bytes/block]:', '?'):>10s}")
assert all(r[1] for r in rows), "a kernel without a name"
open(os.path.join(ROOT, "profiles", "r06_kernel_resources.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(l for l in out if any(k in l for k in ("fused_sv_mvbs_kernel<double, float, true", "drift_kernel<true, true", "sv_noise_fast_kernel<double, true",
                                                       "sv_complex_fft_kernel<float, double, double, 4, false", "mvbs_of_sv_fixed_kernel<double, true", "kernel  "))))
