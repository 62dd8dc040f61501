#!/bin/bash
# pipelined throughput against the number of streams / contexts (and hardware queues): bench.py --streams S
cd $GRAFT_REPO_ROOT
for cfg in "${CONFIGS:-2}"; do
for s in ${STREAMS:-1 2 3 4 6 8 12 16 23 31}; do
  python bench.py --config $cfg --streams $s --steps 400 --warmup 40 --no-cpu-baseline --no-stages --no-h2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config $cfg streams $s'.ljust(28), d['value'], 'scans/s', d['ms_per_step'], 'ms/step  host issue', d.get('host_issue_ms_per_step'))"
done; done
