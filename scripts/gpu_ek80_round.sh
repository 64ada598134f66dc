#!/bin/bash
# one GPU call: EK80 parity tests + cfg4 bench lines (f64 / f32 out) + kernel stats
mkdir -p gpurun_out/ek80
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -x -k "complex or ek80" 2>&1 | tail -15 > gpurun_out/ek80/tests.txt
cat gpurun_out/ek80/tests.txt
python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ek80/bench_f64.json 2> gpurun_out/ek80/bench_f64.err
python bench.py --workload cfg4 --dtype float32 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ek80/bench_f32.json 2> gpurun_out/ek80/bench_f32.err
cat gpurun_out/ek80/bench_f64.json gpurun_out/ek80/bench_f32.json
tail -3 gpurun_out/ek80/bench_f64.err
