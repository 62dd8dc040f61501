#!/bin/bash
# A/B of an environment switch: pipelined throughput + per-layer serial stage times.  usage: ab.sh "VAR=a" "VAR=b" ...
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  env $kv python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('$kv'.ljust(28), d['value'], d['roofline']['gpu_ms_per_step'], 'serial_sum', d['roofline']['stage_ms_sum'])
print('   ', ' '.join(s['stage'].replace('block','b').replace('.0.conv','c').replace('conv','c')+':'+str(round(s['ms']*1000,1)) for s in st))"
done
