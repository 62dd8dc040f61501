#!/bin/bash
# H2D-inclusive and resident rates over the stream count (config 2).  usage: streams_h2d_sweep.sh "3 5 7 9 11 15 23" [bench flags]
cd $GRAFT_REPO_ROOT
for s in $1; do
  timeout -k 10 200 python bench.py --streams $s --steps 400 --warmup 40 --no-cpu-baseline --no-stages ${@:2} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams', d['config']['streams_per_gpu'], 'h2d-inclusive', d['value'], 'resident', d['resident_inputs']['value'], 'arena_MB', d['config']['arena_mb_all_contexts'], 'issue_ms', d['host_issue_ms_per_step'])"
done
