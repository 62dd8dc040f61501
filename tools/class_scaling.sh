#!/bin/bash
# How does ONE kernel class scale with the number of forwards in flight?  -DSPS_DIAG build (tools/ab_build.sh diag -DSPS_DIAG):
# SPS_DIAG_SKIP=1 reuses the coordinate structures of the previous forward (no front-end), SPS_DIAG_SKIP_CLASS drops
# kernel classes (1 coarse 3^4, 2 fine 3^4 (k_conv_px), 4 strided, 8 transposed, 16 conv0); what is left runs on 1..N streams.
# usage (GPU box): bash tools/class_scaling.sh "<skip_class mask>" ...
for m in "$@"; do
  for st in 1 2 3 4 7 11; do
    SPS_LIB=tools/ab/lib_diag.so SPS_DIAG_SKIP=1 SPS_DIAG_SKIP_CLASS=$m python3 bench.py --steps 400 --warmup 40 --streams $st --exact-streams --no-cpu-baseline --no-h2d --no-stages 2>> gpurun_out/cs.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('skip_class', $m, 'streams', $st, 'us/scan', round(1e3 * d['ms_per_step'], 1), 'scans/s', d['value'])"
  done
done
