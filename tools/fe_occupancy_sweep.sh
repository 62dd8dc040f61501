#!/bin/bash
# Occupancy cap of the latency-bound front-end launches (diagnostic build: SPS_FE_LDS_PAD bytes of untouched dynamic LDS per
# workgroup, SPS_FE_LDS_MASK = which of k_points_to_blocks / k_rank_points / k_rank_blocks_rows / k_link_adj / k_maps): pipelined rate.
# usage (GPU box): bash tools/fe_occupancy_sweep.sh <diag tag> "pad[:mask]" ...
cd $GRAFT_REPO_ROOT
tag=$1; shift
export SPS_LIB=tools/ab/lib_$tag.so
for r in 1 2 3; do
for v in "$@"; do
  pad=${v%%:*}; mask=31; [[ "$v" == *:* ]] && mask=${v#*:}
  SPS_FE_LDS_PAD=$pad SPS_FE_LDS_MASK=$mask python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-stages 2>/dev/null | python3 -c "
import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pad $pad mask $mask rep $r: value', d['value'], 'resident', d['resident_value'])"
done; done
