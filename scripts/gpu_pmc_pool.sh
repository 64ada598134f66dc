#!/bin/bash
# PMC counters of the staged value-window pooling kernel (4 x 20000 x 2000; separate passes per counter group)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_pool; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/g$i -o p --output-format csv -- python scripts/perf_pool_value.py 4 20000 2000 > $O/g$i.log 2>&1
  echo "group $i rc $?"; tail -2 $O/g$i.log | cut -c1-200
done
python scripts/pmc_summary.py $O pool_value_mean > $O/summary.csv
cat $O/summary.csv | cut -c1-200
