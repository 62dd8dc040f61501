#!/bin/bash
cd $GRAFT_REPO_ROOT
cp sps_amd/csrc/libsps_hip.so /tmp/libsps_hip.orig.so
for v in NONE SPS_ABLATE_MFMA "SPS_ABLATE_A -DSPS_ABLATE_B -DSPS_ABLATE_STAGE" "SPS_ABLATE_A -DSPS_ABLATE_B -DSPS_ABLATE_STAGE -DSPS_ABLATE_MFMA"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -D$v -o sps_amd/csrc/libsps_hip.so sps_amd/csrc/sps_hip.hip 2>/dev/null
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --streams 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$v'.ljust(70), 'ms/scan', d['ms_per_step'])"
done
cp /tmp/libsps_hip.orig.so sps_amd/csrc/libsps_hip.so
