#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('$1'.ljust(44), d['roofline']['gpu_ms_per_scan'], ' '.join(s['stage'].replace('block','b').replace('.0.conv','c')+':'+str(round(s['ms']*1000)) for s in st if s['stage'][:6] in ('block2','block3','block4','block5','block6')))"; }
run base
for g in "1,2" "1,4" "0,4"; do SPS_GEOM_L2=$g run "L2=$g"; done
for g in "1,4" "1,8" "0,8" "1,2"; do SPS_GEOM_L3=$g run "L3=$g"; done
for g in "1,4" "1,8" "1,16" "0,8"; do SPS_GEOM_L4=$g run "L4=$g"; done
