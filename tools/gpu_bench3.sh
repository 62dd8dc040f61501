#!/bin/bash
# default bench, driver protocol, 23 streams + the engine tests (H2D path).  usage: gpu_bench3.sh <tag>
cd $GRAFT_REPO_ROOT
tag=${1:-b3}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_hip_stream.py -m gpu -q -x -k "engine or predict_cli" > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-stages > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages > $out/bench_driver.json 2> $out/bench_driver.err || { tail -5 $out/bench_driver.err; exit 1; }
timeout -k 10 300 python bench.py --streams 23 --no-cpu-baseline --no-stages > $out/bench_s23.json 2> $out/bench_s23.err || { tail -5 $out/bench_s23.err; exit 1; }
timeout -k 10 300 python bench.py --streams 11 --no-cpu-baseline --no-stages > $out/bench_s11.json 2> $out/bench_s11.err || { tail -5 $out/bench_s11.err; exit 1; }
python - <<PY
import json
for f in ("bench","bench_driver","bench_s11","bench_s23"):
    d=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, 'value', d['value'], 'resident', d['resident_inputs']['value'], 'S', d['config']['streams_per_gpu'], 'arena_all_MB', d['config']['arena_mb_all_contexts'], 'frac', d['roofline']['frac'], 'issue_ms', d['host_issue_ms_per_step'], d['resident_inputs']['host_issue_ms_per_step'])
PY
