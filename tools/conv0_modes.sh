for m in 0 1 2; do
SPS_LIB=tools/ab/lib_diag.so SPS_DIAG_CONV0=$m python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>> gpurun_out/c0.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); st = {s['stage']: s['ms'] * 1000 for s in d['roofline']['stages']}
print('conv0 mode', $m, 'conv0 stage us', round(st['conv0p1s1'], 2), 'maps', round(st['maps'], 2), 'resident', d['resident_value'])"
done
