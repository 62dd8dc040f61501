#!/bin/bash
# One GPU round trip: rocprofv3 kernel trace of a short serial bench run -> per-kernel table + the timeline of one steady-state scan.
# usage: gpu_timeline.sh <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-tl}; shift; o=gpurun_out/$tag; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --streams 1 --steps 60 --warmup 10 --no-cpu-baseline --no-h2d --no-stages "$@" > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 tools/kernel_table.py $o/prof --timeline > $o/kernel_table.txt
rm -rf $o/prof
