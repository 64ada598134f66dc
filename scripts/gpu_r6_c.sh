#!/bin/bash
# round 6: the pipeline helper -- tests, the headline through it, the sharded entry points at world 1 with two streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_pipelined_api.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 900 python bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 2 > $O/head.json 2> $O/head.err
timeout 900 python bench.py --workload cfg5:one --no-cpu-baseline --steps 10 --warmup 2 > $O/one.json 2>> $O/head.err
timeout 900 python bench.py --only-headline --sharded-at-1 --no-cpu-baseline --steps 10 --warmup 2 > $O/sharded2.json 2> $O/sharded.err
timeout 900 python bench.py --only-headline --sharded-at-1 --tile-streams 1 --no-cpu-baseline --steps 10 --warmup 2 > $O/sharded1.json 2>> $O/sharded.err
python - <<'PY'
import json
for f in ("head", "one", "sharded2", "sharded1"):
    try:
        d = json.loads(open(f"gpurun_out/r6c/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    c = d["config"]
    print(f, "Gs/s %.1f" % (d["value"] / 1e9), "ms/pass %.2f" % c["ms_per_pass"], "frac %.3f" % d["roofline"]["frac"], "kernel_ms", d["roofline"].get("kernel_ms"), "each", d["roofline"].get("kernel_ms_each"), "host_ms/call %.3f" % c.get("host_ms_per_call", -1), "streams", c.get("tile_streams"), c.get("ranks", {}).get("backend"))
PY
tail -n 3 $O/head.err; tail -n 3 $O/sharded.err
