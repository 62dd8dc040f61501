"""Probe: do streams of mixed priorities (ROCclr keeps separate hardware-queue pools per priority) raise the pipelined rate?
usage: gpurun -- python tools/prio_probe.py            (prints scans/s for several priority patterns at 7 and 11 streams)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.engine import ScanEngine
from sps_amd.models.models import SPSNet

net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
x = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
Orig = torch.cuda.Stream
for S in (7, 11):
    for name, pat in (("all 0", [0]), ("0,-1", [0, -1]), ("all -1", [-1]), ("0,0,-1", [0, 0, -1]), ("0,-1,1?", [0, -1, 1])):
        it = iter(range(10 ** 9))
        def mk(*a, **k):
            k = dict(k); k["priority"] = pat[next(it) % len(pat)]
            return Orig(*a, **k)
        torch.cuda.Stream = mk
        try:
            eng = ScanEngine(net, streams=S, max_rows=len(x) + 1024, table_rows=64)
        except Exception as e:
            print(S, name, "failed:", str(e)[:80]); torch.cuda.Stream = Orig; continue
        torch.cuda.Stream = Orig
        for _ in range(3 * S):
            eng.reset_table(8); eng.submit(x, 1)
        eng.finish()
        K = 400
        eng.reset_table(K + 8)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            eng.submit(x, 1, row=i % 8)
        eng.finish(); dt = time.perf_counter() - t0
        print(f"S={S:2d} priorities {name:8s}: {K / dt:7.1f} scans/s")
        del eng
