#!/bin/bash
# Stall / cache counters of the LDS-FFT kernels (development aid): separate rocprofv3 --pmc passes over scripts/pmc_hot.py fft
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_fft; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TA_TA_BUSY TD_TD_BUSY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           ; do  # (a GRBM_* group made rocprofv3 abort on this image: left out)
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/g$i -o p --output-format csv -- python scripts/pmc_hot.py fft > $O/g$i.log 2>&1 || tail -3 $O/g$i.log
done
python scripts/pmc_summary.py $O sv_complex_fft > $O/summary.csv
grep -v "true>" $O/summary.csv | cut -d, -f1-3 | cut -c40-200
