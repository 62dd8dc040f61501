"""Probe: does replaying a forward as a HIP graph (captured with torch.cuda.graph around the native launches)
beat 47 individual launches?  Tiny cloud (dispatch floor) and the config-2 cloud, 16 streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
small = torch.from_numpy(synthetic.small_scene(seed=0, n_scan=1500)).cuda()[:, :5].contiguous()
big = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()[:, :5].contiguous()
S = 7
streams = [torch.cuda.Stream() for _ in range(S)]
for name, x in (("tiny", small), ("config2", big)):
    for s in streams:                       # warm-up: arena sizing + weights on every context
        with torch.cuda.stream(s):
            net.model(x); net.model(x)
    torch.cuda.synchronize()
    ref = net.model(x).clone()
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % S]):
            net.model(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K
    graphs, outs = [], []
    for s in streams:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs.append(net.model(x))
        graphs.append(g)
    torch.cuda.synchronize()
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(o, ref) for o in outs)
    t0 = time.perf_counter()
    for i in range(K):
        graphs[i % S].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / K
    # serial latency: one stream
    t0 = time.perf_counter()
    for i in range(100):
        graphs[0].replay()
    torch.cuda.synchronize()
    g1 = (time.perf_counter() - t0) / 100
    with torch.cuda.stream(streams[0]):
        t0 = time.perf_counter()
        for i in range(100):
            net.model(x)
        torch.cuda.synchronize()
    e1 = (time.perf_counter() - t0) / 100
    print(f"{name}: x{S} eager {eager*1e6:.0f} us/scan, graph {graph*1e6:.0f} us/scan (host issue {(t1-t0)/K*1e6:.0f}); "
          f"1 stream eager {e1*1e6:.0f}, graph {g1*1e6:.0f}; graph output identical: {ok}")
