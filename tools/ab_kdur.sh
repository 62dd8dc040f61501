#!/bin/bash
# A/B of private builds by rocprofv3 per-launch kernel durations of the SERIAL forward (no launch gaps, no event noise):
# usage (GPU box): bash tools/ab_kdur.sh tagA tagB ...   -> gpurun_out/kdur_<tag>.json + a side-by-side table (us per launch position)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in "$@"; do
  export SPS_LIB=tools/ab/lib_$t.so
  rm -rf gpurun_out/kdur_prof_$t
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kdur_prof_$t -- python3 bench.py --streams 1 --steps 80 --warmup 10 --no-cpu-baseline --no-h2d --no-stages > /dev/null 2>> gpurun_out/kdur.err || echo "FAILED $t"
  python3 tools/kernel_durations.py gpurun_out/kdur_prof_$t gpurun_out/kdur_$t.json > /dev/null
  rm -rf gpurun_out/kdur_prof_$t
done
python3 - "$@" <<'PY'
import json, sys
tags = sys.argv[1:]
d = {t: json.load(open(f"gpurun_out/kdur_{t}.json")) for t in tags}
n = max(len(v["launches"]) for v in d.values())
print("pos " + " ".join(f"{t:>10s}" for t in tags) + "  kernel")
for i in range(n):
    row = [d[t]["launches"][i] if i < len(d[t]["launches"]) else ("-", float("nan")) for t in tags]
    print(f"{i:3d} " + " ".join(f"{u:10.2f}" for _, u in row) + "  " + row[0][0][:60])
print("sum " + " ".join(f"{d[t]['sum_us']:10.2f}" for t in tags))
PY
