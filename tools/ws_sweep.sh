#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ws_sweep.sh "<flags 1>" "<flags 2>" ...   ("" = the product flags)
# Private build per flag set (loaded through $SPS_LIB), serial stage times of the coarse-level layers + pipelined rate.
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  v="${spec%%;;*}"; envs=""; [[ "$spec" == *";;"* ]] && envs="${spec##*;;}"
  lib=$(mktemp /tmp/libsps_variant.XXXXXX.so)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $lib sps_amd/csrc/sps_hip.hip 2>/dev/null || { echo "[$spec] build failed"; continue; }
  export SPS_LIB=$lib
  env $envs timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
import os; pre=tuple(os.environ.get('SWEEP_STAGES','block2,block3,block4,block5,block6,conv3p,conv4p').split(','))
sel=[s for s in st if s['stage'].startswith(pre)]
print('[$spec]', d['value'], 'scans/s | serial sum', d['roofline']['stage_ms_sum'], '| selected', round(sum(s['ms'] for s in sel)*1000,1), '|', ' '.join(s['stage'].replace('block','b').replace('.0.conv','c').replace('conv','c')+':'+str(round(s['ms']*1000,1)) for s in sel), '| parity', (d.get('parity') or {}).get('max_abs_score_diff'))"
  unset SPS_LIB; rm -f $lib
done
