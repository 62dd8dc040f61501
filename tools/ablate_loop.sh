#!/bin/bash
cd $GRAFT_REPO_ROOT
cp sps_amd/csrc/libsps_hip.so /tmp/libsps_hip.orig.so
for v in "-DNONE" "-DSPS_ABLATE_A -DSPS_ABLATE_B -DSPS_ABLATE_STAGE -DSPS_ABLATE_MFMA" "-DSPS_ABLATE_A -DSPS_ABLATE_B -DSPS_ABLATE_STAGE -DSPS_ABLATE_MFMA -DSPS_ABLATE_LOOP"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o sps_amd/csrc/libsps_hip.so sps_amd/csrc/sps_hip.hip 2>/dev/null
  r=$(timeout 150 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  r1=$(timeout 150 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['gpu_ms_per_scan'])")
  echo "[$v] x23 scans/s, ms/scan: $r | serial $r1"
done
cp /tmp/libsps_hip.orig.so sps_amd/csrc/libsps_hip.so
