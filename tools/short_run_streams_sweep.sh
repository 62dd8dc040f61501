for r in 1 2 3 4 5; do
for s in 4 6 8 3; do
python3 bench.py --steps 20 --warmup 5 --streams $s --exact-streams --no-cpu-baseline --no-stages 2>/dev/null | python3 -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams', $s, 'value', d['value'], 'resident', d['resident_value'])"
done; done
