set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest.log
tail -15 gpurun_out/r03a/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --out gpurun_out/r03a/bench_default.jsonl > gpurun_out/r03a/bench.log 2>&1; echo "bench rc=$?"
tail -c 3000 gpurun_out/r03a/bench.log
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --pings-total 400000 --steps 10 --warmup 2 > gpurun_out/r03a/bench_gloo2.log 2>&1; echo "gloo2 rc=$?"
tail -c 1500 gpurun_out/r03a/bench_gloo2.log
