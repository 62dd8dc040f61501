"""Is the pipelined loop host-bound?  Issue K forwards over 16 streams, record when the host finished issuing
and when the GPU finished executing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
b = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
S = 16
streams = [torch.cuda.Stream() for _ in range(S)]
for i in range(2 * S):
    with torch.cuda.stream(streams[i % S]):
        net(b)
torch.cuda.synchronize()
for K in (64, 400):
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % S]):
            net(b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: host finished issuing after {(t1 - t0) / K * 1e6:.0f} us/scan, GPU finished after {(t2 - t0) / K * 1e6:.0f} us/scan "
          f"(tail after last issue: {(t2 - t1) * 1e6:.0f} us)")
