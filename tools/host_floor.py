"""Host-side floor of the pipelined loop: forwards of a TINY cloud (GPU work negligible) over 16 streams.
The time per forward is what one Python thread needs to issue the ~45 launches of a forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
small = torch.from_numpy(synthetic.small_scene(seed=0, n_scan=1500)).cuda()
streams = [torch.cuda.Stream() for _ in range(16)]
for s in streams:
    with torch.cuda.stream(s):
        net.model(small[:, :5])
torch.cuda.synchronize()
for reserve_big in (False, True):
    if reserve_big:
        big = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
        for s in streams:
            with torch.cuda.stream(s):
                net.model(big[:, :5])
        torch.cuda.synchronize()
    t = time.perf_counter()
    n = 800
    for i in range(n):
        with torch.cuda.stream(streams[i % 16]):
            net.model(small[:, :5])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"tiny cloud, arena sized for {'config 2' if reserve_big else 'the tiny cloud'}: {dt*1e6:.0f} us per forward (host-bound floor)")
