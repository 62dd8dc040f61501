#!/bin/bash
# diagnostic: sweep (G, min waves/SIMD) of the NTW = 1 / 2 conv instantiations on the GPU box
cd $GRAFT_REPO_ROOT
cp sps_amd/csrc/libsps_hip.so /tmp/libsps_hip.orig.so
for cfg in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $cfg -o sps_amd/csrc/libsps_hip.so sps_amd/csrc/sps_hip.hip 2>/dev/null
  s1=$(python bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print(d['roofline']['gpu_ms_per_scan'], ' '.join(s['stage'].replace('block','b').replace('.0.conv','c')+':'+str(round(s['ms']*1000)) for s in st if s['stage'][:5]=='block'))")
  s16=$(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --streams 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$cfg | serial $s1 | x16 $s16"
done
cp /tmp/libsps_hip.orig.so sps_amd/csrc/libsps_hip.so
