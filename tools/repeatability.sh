#!/bin/bash
# Repeatability on ONE box: <reps> alternating runs of the default line and of the driver's protocol.
# usage (GPU box): bash tools/repeatability.sh <out file> [reps]
out=${1:-gpurun_out/repeatability.txt}; reps=${2:-3}
echo "# repeatability on one box: $reps alternating runs of the default line and of the driver's protocol ($(date -u +%F))" > $out
for r in $(seq 1 $reps); do
  python3 bench.py --no-cpu-baseline --no-stages 2>> gpurun_out/repeat.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default 400 steps 8 pipelines: value', d['value'], 'resident', d['resident_value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $out
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages 2>> gpurun_out/repeat.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('driver protocol 20 / 5 (4 pipelines): value', d['value'], 'resident', d['resident_value'], 'ms/step', d['ms_per_step'])" >> $out
done
cat $out
