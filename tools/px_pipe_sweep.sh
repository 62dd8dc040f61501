#!/bin/bash
# pipelined throughput (default bench, 23 streams; configs 2 and 3) and the serial step of private builds:
# usage px_pipe_sweep.sh "<-D flags>" ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$(mktemp /tmp/libsps_variant.XXXXXX.so)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $lib sps_amd/csrc/sps_hip.hip 2>/dev/null || { echo "[$v] build failed"; continue; }
  export SPS_LIB=$lib
  ok=$(timeout -k 10 200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "small_scene or config2 or stress or fused" 2>&1 | tail -1)
  echo "[$v] tests: $ok"
  for cfg in 2 3; do
  timeout -k 10 200 python bench.py --config $cfg --steps 400 --warmup 40 --no-cpu-baseline --no-h2d --no-stages 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] config $cfg:', d['value'], 'scans/s', d['ms_per_step'], 'ms/step')"
  done
  timeout -k 10 200 python bench.py --streams 1 --steps 200 --warmup 20 --no-cpu-baseline --no-h2d --no-stages 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] serial:', d['value'], 'scans/s', d['ms_per_step'], 'ms/step')"
  unset SPS_LIB; rm -f $lib
done
