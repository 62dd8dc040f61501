#!/bin/bash
# Offset-major convolutions (k_om_gemm + k_om_sum) against k_conv, layer by layer, on ONE box: a -DSPS_DIAG build
# (tools/ab_build.sh diag -DSPS_DIAG) is benched under SPS_OM_LAYERS / SPS_OM_RT / SPS_OM_GRID settings.
# usage (GPU box): bash tools/om_sweep.sh "<layers mask>:<rt>:<grid>[:<levels mask>]" ...   -> one line per setting: resident scans/s, serial us, the ten coarse stages
mkdir -p gpurun_out
for cfg in "$@"; do
  IFS=: read -r mask rt grid lv <<< "$cfg"
  SPS_LIB=tools/ab/lib_diag.so SPS_OM=${lv:-28} SPS_OM_LAYERS=$mask SPS_OM_RT=${rt:-2} SPS_OM_GRID=${grid:-768} python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-h2d > gpurun_out/om_${mask}_${rt}_${grid}_${lv}.json 2>> gpurun_out/om.err || { echo "FAILED $cfg"; tail -3 gpurun_out/om.err; continue; }
  python3 - "$cfg" gpurun_out/om_${mask}_${rt}_${grid}_${lv}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = {s["stage"]: s["ms"] * 1000 for s in d["roofline"]["stages"]}
names = ["block2.0.conv1", "block2.0.conv2", "block3.0.conv1", "block3.0.conv2", "block4.0.conv1", "block4.0.conv2", "convtr4p16s2",
         "block5.0.conv1", "block5.0.conv2", "convtr5p8s2", "block6.0.conv1", "block6.0.conv2", "convtr6p4s2"]
row = []
for n in names:
    v = [x for k, x in st.items() if k.split("+")[0] == n]
    row.append(f"{v[0]:5.1f}" if v else "  -  ")
print(f"{sys.argv[1]:>14s} resident {d['resident_value']:7.1f} serial_sum {sum(st.values()):6.1f} conv0 {st.get('conv0p1s1', 0):5.1f} | " + " ".join(row))
PY
done
