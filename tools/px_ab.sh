#!/bin/bash
# A/B of the pair-exact conv kernel: SPS_PX = bit mask of the levels that use k_conv_px (0 = k_conv everywhere).
# usage (GPU box): tools/px_ab.sh [masks...]   prints pipelined scans/s, serial ms and the per-layer serial stage times
cd $GRAFT_REPO_ROOT
for m in "${@:-0 7}"; do
  SPS_PX=$m python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('SPS_PX=$m'.ljust(12), d['value'], d['roofline']['gpu_ms_per_step'], 'serial_sum', d['roofline']['stage_ms_sum'], 'parity', (d.get('parity') or {}).get('max_abs_score_diff'))
print('   ', ' '.join(s['stage'].replace('block','b').replace('.0.conv','c').replace('conv','c')+':'+str(round(s['ms']*1000,1)) for s in st))"
done
