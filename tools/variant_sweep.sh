#!/bin/bash
# usage (on the GPU box, via gpurun): tools/variant_sweep.sh "<flags 1>" "<flags 2>" ...
# builds libsps_hip.so with each set of -D flags, runs the parity tests' core and both bench modes.
cd $GRAFT_REPO_ROOT
cp sps_amd/csrc/libsps_hip.so /tmp/libsps_hip.orig.so
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o sps_amd/csrc/libsps_hip.so sps_amd/csrc/sps_hip.hip 2>/dev/null
  ok=$(python -m pytest tests/test_hip_parity.py -q -m gpu -k "small_scene or config2" 2>&1 | tail -1)
  s1=$(python bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print(d['roofline']['gpu_ms_per_scan'], ' '.join(s['stage'].replace('block','b').replace('.0.conv','c')+':'+str(round(s['ms']*1000)) for s in st if s['stage'][:6] in ('block1','block5','block6','block7','block8')))")
  s16=$(python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "[$v] tests: $ok | x16: $s16 | serial: $s1"
done
cp /tmp/libsps_hip.orig.so sps_amd/csrc/libsps_hip.so
