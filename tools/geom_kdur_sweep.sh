#!/bin/bash
# Launch geometries of the coarse levels (diagnostic build: SPS_GEOM_L<level>="<column tiles per wave>,<splits>"), each measured as
# the serial forward (rocprofv3 kernel durations: sum + the level's launches) and as the pipelined resident-input rate.
# usage (GPU box): bash tools/geom_kdur_sweep.sh <diag tag> "name|ENV=VAL ENV=VAL" ...      (name "base" with an empty list first)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
export SPS_LIB=tools/ab/lib_$tag.so
mkdir -p gpurun_out
for v in "$@"; do
  name=${v%%|*}; envs=${v#*|}
  (
    for e in $envs; do export $e; done
    rm -rf gpurun_out/gk_prof
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gk_prof -- python3 bench.py --streams 1 --steps 60 --warmup 10 --no-cpu-baseline --no-h2d --no-stages > /dev/null 2>> gpurun_out/gk.err || echo "FAILED $name"
    python3 tools/kernel_durations.py gpurun_out/gk_prof gpurun_out/gk_$name.json > /dev/null
    rm -rf gpurun_out/gk_prof
    r1=$(python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-stages --no-h2d 2>> gpurun_out/gk.err | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])")
    r2=$(python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-stages --no-h2d 2>> gpurun_out/gk.err | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])")
    python3 - "$name" "$r1" "$r2" <<'PY'
import json, sys
name, r1, r2 = sys.argv[1:4]
d = json.load(open(f"gpurun_out/gk_{name}.json"))
us = [u for _, u in d["launches"]]
print(f"{name:28s} serial sum {d['sum_us']:7.2f} us  launches {len(us)}  pos 12-24: " + " ".join(f"{u:5.1f}" for u in us[12:24]) + f"   resident {r1} {r2}")
PY
  )
done
