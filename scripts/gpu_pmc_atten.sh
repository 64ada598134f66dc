#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_atten; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python scripts/pmc_atten.py > $O/kt.log 2>&1
grep -h attenuated $O/kt/*kernel_stats.csv $O/kt/*/*kernel_stats.csv 2>/dev/null | cut -c1-60,100-260
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/$n -o p --output-format csv -- python scripts/pmc_atten.py > $O/$n.log 2>&1
done
python scripts/pmc_summary.py $O attenuated_slide attenuated_limits > $O/summary.csv
cat $O/summary.csv | cut -c1-200
