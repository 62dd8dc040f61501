"""Throughput when B scans share one forward (batch column, collate_fn layout) vs one scan per forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
scenes = [synthetic.make_scene(scan_seed=1 + i, batch_index=i)["batch"] for i in range(8)]
S = 16
streams = [torch.cuda.Stream() for _ in range(S)]
for B in (1, 2, 4, 8):
    x = torch.from_numpy(np.concatenate(scenes[:B], 0)).cuda()
    for s in streams:
        with torch.cuda.stream(s):
            net(x)
    torch.cuda.synchronize()
    K = max(32, 320 // B)
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % S]):
            net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    with torch.cuda.stream(streams[0]):
        t0 = time.perf_counter()
        for i in range(20):
            net(x)
        torch.cuda.synchronize()
    d1 = (time.perf_counter() - t0) / 20
    print(f"batch {B}: {len(x)} rows/forward; x16 streams {dt*1e3:.3f} ms/forward = {B/dt:.0f} scans/s; 1 stream {d1*1e3:.3f} ms/forward = {B/d1:.0f} scans/s")
