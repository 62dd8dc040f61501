#!/bin/bash
# A/B on ONE box: bench.py for every tools/ab/lib_<tag>.so given, alternating, <reps> times.
# usage (GPU box): bash tools/ab_bench.sh <reps> "<bench args>" tagA tagB ...   -> gpurun_out/ab_<tag>_<rep>.json, summary on stdout
reps=$1; args=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for t in "$@"; do
    [ -f tools/ab/lib_$t.so ] || { echo "SKIPPED $t: tools/ab/lib_$t.so does not exist (tools/ab_build.sh $t ...)"; continue; }
    SPS_LIB=tools/ab/lib_$t.so python3 bench.py $args > gpurun_out/ab_${t}_$r.json 2>> gpurun_out/ab.err || echo "FAILED $t $r"
  done
done
python3 - "$reps" "$@" <<'PY'
import json, sys
reps = int(sys.argv[1]); tags = sys.argv[2:]
for t in tags:
    v, rv, ser = [], [], []
    for r in range(1, reps + 1):
        try:
            d = json.loads(open(f"gpurun_out/ab_{t}_{r}.json").read().strip().splitlines()[-1])
        except Exception as e:
            print(t, r, "unreadable", e); continue
        v.append(d["value"]); rv.append(d["resident_value"]); ser.append(d["roofline"].get("serial_kernel_us_per_step"))
    print(f"{t:12s} value {v}  resident {rv}  serial_us {ser}")
PY
