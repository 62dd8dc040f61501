"""How long does the HOST need to issue one forward (async launches only)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet, get_context
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
b = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
for _ in range(5): net(b)
torch.cuda.synchronize()
ctx = get_context(0)
scores = torch.empty(len(b), device="cuda")
st = torch.cuda.current_stream().cuda_stream
for label, fn in (("SPSNet.forward", lambda: net(b)),
                  ("ctx.forward (ctypes only)", lambda: ctx.forward(b.data_ptr(), 6, len(b), 0.1, scores.data_ptr(), st))):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30): fn()
    t_issue = (time.perf_counter() - t) / 30
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t) / 30
    print(f"{label}: host issue {t_issue*1e6:.0f} us/scan, end-to-end {t_total*1e6:.0f} us/scan")
