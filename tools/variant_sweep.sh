#!/bin/bash
# usage (on the GPU box, via gpurun): tools/variant_sweep.sh "<flags 1>" "<flags 2>" ...
# Builds a PRIVATE copy of the library with each set of -D flags (the product libsps_hip.so is never touched: the
# bindings load $SPS_LIB instead), runs the core parity tests and both bench modes against it.
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$(mktemp /tmp/libsps_variant.XXXXXX.so)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $lib sps_amd/csrc/sps_hip.hip 2>/dev/null || { echo "[$v] build failed"; continue; }
  export SPS_LIB=$lib
  ok=$(python -m pytest tests/test_hip_parity.py -q -m gpu -k "small_scene or config2" 2>&1 | tail -1)
  s1=$(python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print(d['roofline']['gpu_ms_per_step'], ' '.join(s['stage'].replace('block','b').replace('.0.conv','c')+':'+str(round(s['ms']*1000)) for s in st if s['stage'][:6] in ('block1','block5','block6','block7','block8')))")
  sp=$(python bench.py --no-cpu-baseline --no-stages --no-h2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "[$v] tests: $ok | pipelined: $sp scans/s | serial: $s1"
  unset SPS_LIB; rm -f $lib
done
