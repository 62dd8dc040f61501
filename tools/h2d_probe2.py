"""Where the host time of the deferred-issue H2D path goes (DIAGNOSTIC): the engine's own loop with per-operation timers."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sps_amd import synthetic
from sps_amd.engine import ScanEngine
from sps_amd.models.models import SPSNet
dev = torch.device("cuda", 0)
net = bench.synthetic_weights(SPSNet(bench.CFG)).to(dev).eval().freeze()
b = synthetic.make_scene(scan_seed=1)["batch"]
pinned = [torch.from_numpy(b).pin_memory() for _ in range(4)]
K = 300
import platform, os
print('host', platform.node(), 'cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for S in (7,):
    eng = ScanEngine(net, dev, streams=S, max_rows=len(b), table_rows=K, stage_cols=6)
    T = {}
    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
        return w
    eng._to_device = timed("to_device", eng._to_device)
    ev_sync, ev_rec = torch.cuda.Event.synchronize, torch.cuda.Event.record
    torch.cuda.Event.synchronize = timed("event.synchronize", ev_sync)
    torch.cuda.Event.record = timed("event.record", ev_rec)
    tcopy = torch.Tensor.copy_
    torch.Tensor.copy_ = timed("copy_", tcopy)
    tpin = torch.Tensor.is_pinned
    torch.Tensor.is_pinned = timed("is_pinned", tpin)
    fm = net.forward_metrics
    net.forward_metrics = timed("forward_metrics", fm)
    for rep in range(2):
        T.clear()
        eng.reset_table(K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            eng.submit(pinned[i % 4], 1, row=i)
        issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        eng.finish()
        print(f"S={S} rep {rep}: issue {issue/K*1e3:.3f} ms/step total {tot/K*1e3:.3f} ms/step -> {K/tot:.0f} scans/s | " +
              " ".join(f"{k} {v/K*1e3:.3f}" for k, v in T.items()))
    torch.cuda.Event.synchronize = ev_sync; torch.cuda.Event.record = ev_rec; torch.Tensor.copy_ = tcopy; torch.Tensor.is_pinned = tpin
