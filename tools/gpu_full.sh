#!/bin/bash
# One GPU round trip: the whole GPU test suite, then the default bench, the driver protocol and (optionally) more.  usage: gpu_full.sh <tag>
cd $GRAFT_REPO_ROOT
tag=${1:-full}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest.log 2>&1; rc=$?
tail -25 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err || { tail -5 $out/bench_driver.err; exit 1; }
timeout -k 10 300 python bench.py --streams 23 --no-cpu-baseline --no-stages > $out/bench_s23.json 2> $out/bench_s23.err || { tail -5 $out/bench_s23.err; exit 1; }
python - <<PY
import json
for f in ("bench","bench_driver","bench_s23"):
    d=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, 'value', d['value'], 'resident', d['resident_inputs']['value'], 'S', d['config']['streams_per_gpu'], 'arena_all_MB', d['config']['arena_mb_all_contexts'], 'frac', d['roofline']['frac'], 'issue_ms', d['host_issue_ms_per_step'], d['resident_inputs']['host_issue_ms_per_step'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    if d.get('parity'): print('  parity', d['parity']['max_abs_score_err_vs_oracle'], d['parity']['label_mismatches_outside_1e-5_band'])
PY
