"""Training-step timing on the MI355X: train-mode forward + native backward + Adam on one config-2-size scan
(SPSNet.training_step, scripts/train.py's inner loop).  usage: gpurun -- python tools/train_timing.py [--azimuth 1750]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet

ap = argparse.ArgumentParser()
ap.add_argument("--azimuth", type=int, default=1750)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=1, help="scans per training step (the reference's config.yaml trains with BATCH_SIZE 2)")
args = ap.parse_args()
cfg = dict(bench.CFG)
cfg["TRAIN"] = {"LR": 7e-5, "WEIGHT_DECAY": 1e-4, "LR_EPOCH": 1, "LR_DECAY": 0.99}
torch.manual_seed(0)
net = bench.synthetic_weights(SPSNet(cfg)).cuda().train()
(opt,), _ = net.configure_optimizers()
batch = torch.from_numpy(synthetic.collate([synthetic.make_scene(scan_seed=1 + b, x_offset=3.0 * b, n_azimuth=args.azimuth)["batch"]
                                            for b in range(args.batch)])).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    out = net.training_step(batch, 0)
    out["loss"].backward()
    opt.step()
    return out["loss"]
for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(args.steps):
    loss = step()
host = (time.perf_counter() - t) / args.steps          # time to ISSUE a step (the loop never waits for the GPU)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / args.steps
# forward only / backward only
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
opt.zero_grad(set_to_none=True)
e[0].record(); out = net.training_step(batch, 0); e[1].record(); out["loss"].backward(); e[2].record()
torch.cuda.synchronize()
print(f"rows {len(batch)}: {dt * 1e3:.2f} ms per training step ({1 / dt:.1f} steps/s), loss {float(loss.detach()):.4f}; "
      f"GPU forward {e[0].elapsed_time(e[1]):.2f} ms, backward {e[1].elapsed_time(e[2]):.2f} ms; host issue {host * 1e3:.2f} ms per step")
