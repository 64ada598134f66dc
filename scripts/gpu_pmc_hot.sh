#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_hot; rm -rf $O; mkdir -p $O
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/$n -o p --output-format csv -- python scripts/pmc_hot.py ${1:-all} > $O/$n.log 2>&1
done
python scripts/pmc_summary.py $O fast_kernel uniform_kernel drift_kernel fused_sv_mvbs_kernel mvbs_of_sv sv_complex_fft > $O/summary.csv
python - <<'PY'
import csv, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open("gpurun_out/pmc_hot/summary.csv")):
    rows[r["kernel"]][r["Counter_Name"]] = float(r["mean_value"]); rows[r["kernel"]]["vgpr"] = r["vgpr"]
for k, v in rows.items():
    if "true>" in k and "fft" in k: continue
    n = 800e6 if "fft" not in k else 81.92e6
    # VALU-issue floor: wavefront instructions x 4 cycles over 256 CUs x 4 SIMDs at 2.4 GHz (what the kernel would take
    # if nothing but VALU issue limited it)
    print("%-70s VALU/sample %.1f SALU/sample %.1f LDS/sample %.2f  VALU-issue floor %.2f ms  VGPRs per lane %s (rocprofv3's Arch_VGPR x 2: it counts pairs on wave64)" % (
        k[-70:], v.get("SQ_INSTS_VALU", 0) * 64 / n, v.get("SQ_INSTS_SALU", 0) * 64 / n, v.get("SQ_INSTS_LDS", 0) * 64 / n,
        v.get("SQ_INSTS_VALU", 0) * 4 / 1024 / 2.4e9 * 1e3, 2 * int(float(v["vgpr"]))))
PY
