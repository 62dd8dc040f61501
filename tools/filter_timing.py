"""Online path timing (sps_node.callback minus ROS): StableFilter on config-2-size scans -- per-scan latency with one scan in
flight (the node's T / P / I log line) and throughput with several scans in flight (submit() without waiting)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet
from sps_amd.pipeline import StableFilter

torch.manual_seed(0)
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
map_pts = synthetic.build_map()
scans = [synthetic.lidar_scan(200 + i, x_offset=0.5 * i)[:, :3].astype(np.float32) for i in range(8)]
ang = 0.1
pose = np.array([[np.cos(ang), -np.sin(ang), 0, 0.3], [np.sin(ang), np.cos(ang), 0, -0.2], [0, 0, 1, 0.0], [0, 0, 0, 1.0]])
f = StableFilter(net, torch.from_numpy(map_pts), voxel_size=0.1, epsilon=0.84)
dev_scans = [torch.from_numpy(s).cuda() for s in scans]
for s in dev_scans[:3]:
    f(s, pose)
torch.cuda.synchronize()
rs = [f(s, pose) for s in dev_scans * 4]
lat = np.array([[r.t_total, r.t_prune, r.t_infer] for r in rs]) * 1e3
print(f"one scan in flight ({len(scans[0])} points, map {len(map_pts)} points): total {lat[:,0].mean():.3f} ms per scan "
      f"(P: transform + submap {lat[:,1].mean():.3f} ms, I: forward + filter {lat[:,2].mean():.3f} ms GPU) -> {1e3 / lat[:,0].mean():.0f} Hz; "
      f"S = {rs[0].n_scan_voxels} scan voxels, M = {rs[0].n_submap_voxels} submap voxels, kept {len(rs[0].filtered)} of {len(scans[0])}")
torch.cuda.synchronize()
t = time.perf_counter()
pend = [f.submit(s, pose) for s in dev_scans * 8]
out = [p.result() for p in pend]
dt = time.perf_counter() - t
print(f"{len(pend)} scans issued back to back on one stream, one synchronisation each at the end: {len(pend) / dt:.0f} scans/s")
hs = [torch.from_numpy(s).pin_memory() for s in scans]
t = time.perf_counter()
pend = [f.submit(s, pose) for s in hs * 8]
out = [p.result() for p in pend]
dt = time.perf_counter() - t
print(f"same, scans arriving in pinned host memory: {len(pend) / dt:.0f} scans/s")
