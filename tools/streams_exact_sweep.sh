for st in 2 3 4 5 6 7 8 11; do
python3 bench.py --steps 400 --warmup 40 --streams $st --exact-streams --no-cpu-baseline --no-stages 2>> gpurun_out/ss.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams', $st, 'value(h2d)', d['value'], 'resident', d['resident_value'], 'us/scan', round(1e3 * d['ms_per_step'], 1))"
done
