#!/bin/bash
# Refresh profiles/traffic.json for the CURRENT build of the kernels (two separate PMC passes, MI355X_MICROARCH.md HBM section).
# usage: gpurun -- bash tools/collect_traffic.sh   (then copy gpurun_out/traffic/traffic.json to profiles/traffic.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/traffic; mkdir -p $o
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $o/pmc_$ctr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-h2d --streams 1 > /dev/null 2> $o/err_$ctr.txt
done
python3 tools/traffic_pmc.py $o/pmc_FETCH_SIZE $o/pmc_WRITE_SIZE $o/traffic.json
rm -rf $o/pmc_FETCH_SIZE $o/pmc_WRITE_SIZE
