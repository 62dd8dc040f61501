#!/bin/bash
# full forward: streams x hardware queues (GPU_MAX_HW_QUEUES) [x extra bench flags].  usage: bash tools/queues_sweep.sh "S:Q[:flags]" ...
for cfg in "$@"; do
  IFS=: read -r st q fl <<< "$cfg"
  GPU_MAX_HW_QUEUES=$q python3 bench.py --steps ${STEPS:-400} --warmup ${WARMUP:-40} --streams $st --exact-streams --no-cpu-baseline --no-stages $fl 2>> gpurun_out/qs.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams', $st, 'queues', $q, '$fl', 'value', d['value'], 'resident', d['resident_value'], 'us/scan', round(1e3 * d['ms_per_step'], 1))"
done
