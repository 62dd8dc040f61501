"""Host-side profile of the training step (cProfile, top cumulative entries).  usage: gpurun -- python tools/train_host_profile.py"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet

cfg = dict(bench.CFG)
cfg["TRAIN"] = {"LR": 7e-5, "WEIGHT_DECAY": 1e-4, "LR_EPOCH": 1, "LR_DECAY": 0.99}
torch.manual_seed(0)
net = bench.synthetic_weights(SPSNet(cfg)).cuda().train()
(opt,), _ = net.configure_optimizers()
batch = torch.from_numpy(synthetic.make_scene(scan_seed=1, n_azimuth=1750)["batch"]).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    out = net.training_step(batch, 0)
    out["loss"].backward()
    opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
