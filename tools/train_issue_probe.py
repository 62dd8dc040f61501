import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet
cfg = dict(bench.CFG); cfg["TRAIN"] = {"LR": 7e-5, "WEIGHT_DECAY": 1e-4, "LR_EPOCH": 1, "LR_DECAY": 0.99}
torch.manual_seed(0)
net = bench.synthetic_weights(SPSNet(cfg)).cuda().train()
(opt,), _ = net.configure_optimizers()
batch = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
def step(sync):
    ts = []
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); opt.zero_grad(set_to_none=True); ts.append(time.perf_counter() - t)
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); out = net.training_step(batch, 0); ts.append(time.perf_counter() - t)
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); out["loss"].backward(); ts.append(time.perf_counter() - t)
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); opt.step(); ts.append(time.perf_counter() - t)
    return ts
for _ in range(5): step(True)
import numpy as np
for sync in (True, False):
    acc = np.zeros(4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): acc += step(sync)
    tot = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("sync between phases" if sync else "free running      ", "zero_grad %.3f  forward %.3f  backward %.3f  opt.step %.3f ms (host time per call); loop %.3f ms/step" % (*(acc / 40 * 1e3), tot / 40 * 1e3))
