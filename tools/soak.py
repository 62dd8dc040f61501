"""Soak: thousands of forwards over many streams with alternating cloud sizes; outputs must stay bit-identical
and device memory must not grow."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet, get_context
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
clouds = [torch.from_numpy(synthetic.make_scene(scan_seed=1 + i, n_azimuth=az)["batch"]).cuda() for i, az in enumerate((1750, 400, 1000))]
S = 23
streams = [torch.cuda.Stream() for _ in range(S)]
ref = []
for c in clouds:
    ref.append(net(c).clone())
torch.cuda.synchronize()
for s in streams:
    with torch.cuda.stream(s):
        for c in clouds:
            net(c)
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
t0 = time.perf_counter()
N = 6000
bad = 0
keep = []
for i in range(N):
    j = i % 3
    with torch.cuda.stream(streams[i % S]):
        out = net(clouds[j])
        if i % 97 == 0:
            keep.append((j, out))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for j, out in keep:
    bad += int(not torch.equal(out, ref[j]))
for s in streams:
    get_context(0, s.cuda_stream).check_errors(s.cuda_stream)
free1, _ = torch.cuda.mem_get_info()
print(f"{N} forwards in {dt:.2f} s ({N/dt:.0f}/s, mixed sizes {[len(c) for c in clouds]}); {len(keep)} sampled outputs, {bad} differ from the "
      f"first run; free device memory before/after: {free0/2**30:.2f} / {free1/2**30:.2f} GiB")
assert bad == 0 and free0 - free1 < 64 * 2**20
