#!/bin/bash
# usage (GPU box): tools/px_sweep.sh "<-D flags 1>" "<-D flags 2>" ...   private build per flag set (never the product .so),
# core parity tests, then pipelined scans/s + per-layer serial stage times with SPS_PX taken from the environment.
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$(mktemp /tmp/libsps_variant.XXXXXX.so)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $lib sps_amd/csrc/sps_hip.hip 2>/dev/null || { echo "[$v] build failed"; continue; }
  export SPS_LIB=$lib
  ok=$(timeout -k 10 200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "small_scene or config2 or stress or fused" 2>&1 | tail -1)
  echo "[$v] tests: $ok"
  timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('   ', d['value'], 'scans/s', d['roofline']['gpu_ms_per_step'], 'ms  serial_sum', d['roofline']['stage_ms_sum'])
print('   ', ' '.join(s['stage'].replace('block','b').replace('.0.conv','c').replace('conv','c')+':'+str(round(s['ms']*1000,1)) for s in st if (s['stage'][:5] in ('maps', 'block')) or '$ALL_STAGES'))"
  unset SPS_LIB; rm -f $lib
done
