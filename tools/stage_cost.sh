#!/bin/bash
# diagnostic: pipelined (16-stream) and serial cost of the front-end vs the convolutions
cd $GRAFT_REPO_ROOT
for S in 1 16; do for k in 0 1; do SPS_DIAG_SKIP=$k python bench.py --steps 300 --warmup 30 --no-cpu-baseline --streams $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $S skip $k (1=front-end reused, 2=no convs):', d['ms_per_step'], 'ms/scan')"; done; done
