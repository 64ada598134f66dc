#!/bin/bash
# one GPU call: EK80 parity tests + cfg4 bench lines (f64 / f32 out) + kernel stats + PMC of the FFT kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ek80; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -q -k "complex or ek80 or EK80" 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
if [ "$1" != "notime" ]; then
python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_f64.json 2> $O/bench_f64.err
python bench.py --workload cfg4 --dtype float32 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
cat $O/bench_f64.json $O/bench_f32.json
fi
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python scripts/perf_ek80.py 2 20000 8192 4 > $O/probe.txt 2>&1
grep -E "^BB|^CW" $O/probe.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE WRITE_SIZE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $O/pmc_$n -o p --output-format csv -- python scripts/perf_ek80.py 2 20000 8192 4 > $O/pmc_$n.log 2>&1
done
python scripts/pmc_summary.py $O sv_complex > $O/pmc_summary.csv
grep fft $O/pmc_summary.csv | cut -c1-200
