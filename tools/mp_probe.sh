cd $GRAFT_REPO_ROOT
export SPS_LIB=tools/ab/lib_diag.so SPS_OM=0
for np in 1 2 3; do
for st in 4 7; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 2953$np bench.py --gpus $np --backend gloo --steps 400 --warmup 40 --streams $st --no-cpu-baseline --no-stages --no-h2d --force-dist 2>> gpurun_out/mp.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ranks', d['n_gpus'], 'streams', d['config'].get('streams_per_gpu'), 'value', d['value'], 'resident', d.get('resident_value'), 'ms/step', d['ms_per_step'])
"
done
done
