#!/bin/bash
# usage: gpurun_pmc.sh <tag> <counter list...>   (one rocprofv3 pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --streams 1 > /dev/null 2> gpurun_out/pmc_$tag.err
