#!/bin/bash
# One GPU round trip: parity tests, then the bench (pipelined + per-layer serial stage times).  usage: gpu_check.sh <tag> [pytest args]
cd $GRAFT_REPO_ROOT
tag=${1:-run}; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q "$@" > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$out/bench.json")); st=d['roofline']['stages']
print('value', d['value'], 'gpu_ms', d['roofline']['gpu_ms_per_step'], 'serial_sum', d['roofline']['stage_ms_sum'], 'h2d', d.get('h2d_inclusive'))
print(' '.join(s['stage'].replace('block','b').replace('.0.conv','c').replace('conv','c')+':'+str(round(s['ms']*1000,1)) for s in st))
print('parity', d.get('parity'))
PY
