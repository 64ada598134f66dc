#!/bin/bash
# Instruction / stall counters of the staged value-window kernel (development aid): separate rocprofv3 --pmc passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_vw; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/g$i -o p --output-format csv -- python scripts/perf_value_windows.py 4 20000 2000 ${1:-f64} > $O/g$i.log 2>&1 || tail -3 $O/g$i.log
done
python scripts/pmc_summary.py $O pool_value_mean_staged > $O/summary.csv
cut -d, -f1-3 $O/summary.csv | cut -c1-200
