# usage: bash tools/ab_stage.sh <reps> <stage> tagA tagB ...   per-stage time + resident rate for A/B libs, alternating on one box
reps=$1; stage=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for t in "$@"; do
    SPS_LIB=tools/ab/lib_$t.so python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>> gpurun_out/abs.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); st = {s['stage']: s['ms'] * 1000 for s in d['roofline']['stages']}
print('$t', 'rep', $r, '$stage', round(st['$stage'], 2), 'us  resident', d['resident_value'], ' value', d['value'])"
  done
done
