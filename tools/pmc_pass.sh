#!/bin/bash
# one rocprofv3 counter pass over a short serial bench run, bounded in time: pmc_pass.sh <tag> <counters...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
echo "pass $tag: $*"
timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-h2d --streams 1 > /dev/null 2> gpurun_out/pmc_$tag.err
echo "pass $tag: exit $?"
