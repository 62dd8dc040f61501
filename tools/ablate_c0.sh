#!/bin/bash
cd $GRAFT_REPO_ROOT
cp sps_amd/csrc/libsps_hip.so /tmp/libsps_hip.orig.so
for v in NONE SPS_ABLATE_C0FETCH; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -D$v -o sps_amd/csrc/libsps_hip.so sps_amd/csrc/sps_hip.hip 2>/dev/null
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('$v'.ljust(30), d['roofline']['gpu_ms_per_scan'], ' '.join(s['stage']+':'+str(round(s['ms']*1000,1)) for s in st[:6]))"
done
cp /tmp/libsps_hip.orig.so sps_amd/csrc/libsps_hip.so
