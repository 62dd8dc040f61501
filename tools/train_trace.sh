#!/bin/bash
# per-dispatch view of one training step under rocprofv3: tools/train_trace.sh <tag> [env assignments...]  ->  gpurun_out/train_trace_<tag>.txt
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tt_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt_$tag -- python3 tools/train_timing.py --steps 4 > gpurun_out/tt_$tag.log 2>&1 || exit 1
python3 tools/train_trace.py $(ls gpurun_out/tt_$tag/*/*kernel_trace.csv | head -1) > gpurun_out/train_trace_$tag.txt 2>&1
rm -rf gpurun_out/tt_$tag
grep "k_wgrad<" gpurun_out/train_trace_$tag.txt | awk '{print $1,$2,$3}' | head -31 > gpurun_out/wg_$tag.txt
tail -1 gpurun_out/train_trace_$tag.txt
