#!/bin/bash
# One GPU round trip: rocprofv3 kernel stats of a short serial bench run -> per-kernel table.  usage: gpu_ktable.sh <tag> [n rows]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-kt}; o=gpurun_out/$tag; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --streams 1 --steps 60 --warmup 10 --no-cpu-baseline --no-h2d --no-stages > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python3 tools/kernel_table.py $o/prof > $o/kernel_table.txt
rm -rf $o/prof
head -${2:-34} $o/kernel_table.txt | cut -c1-64,65-110
