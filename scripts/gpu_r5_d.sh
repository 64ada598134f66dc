#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
# same box: the row-major walk of walk3 (6.1 TB/s on two boxes) against walk4's NT = 256 line (5.2 on a third)
./scripts/probes/bin/hbm_walk3b_probe | head -4 > $O/walk_same_box.txt; ./scripts/probes/bin/hbm_walk4_probe | head -2 >> $O/walk_same_box.txt; cat $O/walk_same_box.txt
python scripts/perf_shard_host_profile.py > $O/shard_host_profile.txt 2>&1; grep -E "^==|tottime|ncalls" -A0 $O/shard_host_profile.txt | head; grep -A32 "sharded=True" $O/shard_host_profile.txt | head -45
python -m pytest tests/test_gpu_multi_rank.py tests/test_gpu_sharded_sonars.py tests/test_gpu_sharded.py -m gpu -q --maxfail=10 > $O/tests_shard.txt 2>&1; echo "shard tests rc $?"; tail -4 $O/tests_shard.txt
python bench.py --gpus 2 --backend gloo --single-device --no-cpu-baseline --steps 5 --warmup 2 2>$O/err_gloo2.txt | tail -1 > $O/bench_gloo2_full_tiles.json
python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3 --sharded-at-1 2>$O/err_head_sharded.txt | tail -1 > $O/bench_head_sharded_at_1.json
python - <<'PY'
import json
for f in ("bench_head_sharded_at_1", "bench_gloo2_full_tiles"):
    try:
        d = json.loads(open(f"gpurun_out/r5d/{f}.json").read())
        c = d["config"]
        print(f, "| %.1f G/s  %.2f ms/pass  ops-level %.2f ms/pass  kernel %.2f ms  frac %.3f  host %.3f ms/call" % (
            d["value"] / 1e9, c["ms_per_pass"], c["ops_level_ms_per_pass"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], c["host_ms_per_call"]))
    except Exception as e:
        print(f, "FAILED", repr(e)[:200])
PY
grep -v amdgpu.ids $O/err_gloo2.txt | grep -A8 "rank0.*Traceback" | head -20
