#!/bin/bash
# Front-end ablation table: -DSPS_DIAG -DSPS_ABLATE_FE=<bits> builds (tools/ab_build.sh fe<bits> ...), each run
# under rocprofv3 --kernel-trace --stats on a short serial bench; prints the average duration of the five front-end kernels.
# Ablated builds compute WRONG maps (that is the point: what a piece costs); only the ablated kernel's own column means something.
# usage (GPU box): bash tools/fe_ablation.sh 0 1 2 4 8 16 24 32 64 96 128
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fe
printf "%-6s %12s %12s %12s %12s %12s\n" bits k_points_to_blocks k_rank_points k_rank_blocks_rows k_link_adj k_maps
for b in "$@"; do
  lib=tools/ab/lib_fe$b.so
  [ -f $lib ] || { echo "fe$b: $lib missing"; continue; }
  rm -rf gpurun_out/fe/p$b
  SPS_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fe/p$b -- python3 bench.py --streams 1 --steps 40 --warmup 10 --no-cpu-baseline --no-h2d --no-stages > gpurun_out/fe/b$b.json 2> gpurun_out/fe/b$b.err || { echo "fe$b failed"; tail -2 gpurun_out/fe/b$b.err; continue; }
  python3 - gpurun_out/fe/p$b $b <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])))
def us(name):
    r = [x for x in rows if name in x["Name"]]
    return f"{float(r[0]['AverageNs']) / 1000:12.2f}" if r else " " * 12
print(f"{sys.argv[2]:<6s} " + " ".join(us(k) for k in ("k_points_to_blocks", "k_rank_points", "k_rank_blocks_rows", "k_link_adj", "k_maps")))
PY
  rm -rf gpurun_out/fe/p$b
done
