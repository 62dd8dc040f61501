#!/bin/bash
# diagnostic: build conv-kernel ablation variants ON THE GPU BOX and time the stages (outputs are wrong by design)
cd $GRAFT_REPO_ROOT
cp sps_amd/csrc/libsps_hip.so /tmp/libsps_hip.orig.so
for v in NONE SPS_ABLATE_MFMA SPS_ABLATE_A SPS_ABLATE_B SPS_ABLATE_STAGE "SPS_ABLATE_A -DSPS_ABLATE_B" "SPS_ABLATE_A -DSPS_ABLATE_B -DSPS_ABLATE_STAGE" "SPS_ABLATE_A -DSPS_ABLATE_B -DSPS_ABLATE_STAGE -DSPS_ABLATE_MFMA"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -D$v -o sps_amd/csrc/libsps_hip.so sps_amd/csrc/sps_hip.hip 2>/dev/null
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('$v'.ljust(70), d['roofline']['gpu_ms_per_scan'], ' '.join(s['stage'].replace('block','b').replace('.0.conv','c')+':'+str(round(s['ms']*1000)) for s in st if s['stage'][:6] in ('block1','block5','block7','block8','convtr')))"
done
cp /tmp/libsps_hip.orig.so sps_amd/csrc/libsps_hip.so
