#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_pmc_depth; rm -rf $O; mkdir -p $O
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_INSTS_BRANCH"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/$n -o p --output-format csv -- python scripts/perf_depth_fused.py 100000 > $O/$n.log 2>&1
done
python scripts/pmc_summary.py $O fused_sv_mvbs_kernel > $O/summary.csv
python - <<'PY'
import csv, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open("gpurun_out/r6_pmc_depth/summary.csv")):
    rows[r["kernel"]][r["Counter_Name"]] = float(r["mean_value"]); rows[r["kernel"]]["meta"] = (r["vgpr"], r["lds"], r["scratch"])
names = sorted({c for v in rows.values() for c in v if c != "meta"})
n = 800e6
for k, v in rows.items():
    print(k[-40:], v["meta"])
    print("   " + "  ".join(f"{c}={v.get(c, 0) * 64 / n:.2f}/sample" if "INSTS" in c else f"{c}={v.get(c, 0):.3g}" for c in names))
PY
find $O -name "*.csv" -size +1M -delete
