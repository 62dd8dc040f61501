"""Variant-A submap: scipy cKDTree.query_ball_tree (what the reference does per scan in a DataLoader worker,
blt_dataset.py:224-226,258-271) vs the device radius grid; variant-B (util.prune) on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial import cKDTree
from sps_amd import synthetic
import sps.datasets.blt_dataset as blt
import sps.datasets.util as util
map_pts = synthetic.build_map().astype(np.float64)
scan = synthetic.lidar_scan(1).astype(np.float64)
print("map points", len(map_pts), "scan points", len(scan))
tree = cKDTree(map_pts[:, :3])
t = time.time(); hits = cKDTree(scan[:, :3]).query_ball_tree(tree, 0.1); idx = np.concatenate([np.asarray(h, dtype=np.int64) for h in hits]); t_scipy = time.time() - t
print(f"scipy query_ball_tree: {t_scipy*1e3:.1f} ms, {len(idx)} hits")
t = time.time(); sub = blt.DeviceRadiusSubmap(map_pts[:, :3], 0.1); torch.cuda.synchronize(); print(f"device grid build (once per map): {(time.time()-t)*1e3:.1f} ms")
for _ in range(3): out, cnt = sub.query(scan[:, :3])
torch.cuda.synchronize()
s_dev = torch.from_numpy(scan[:, :3]).cuda()
t = time.time()
for _ in range(20): out, cnt = sub.query(s_dev)
torch.cuda.synchronize(); t_dev = (time.time() - t) / 20
print(f"device radius submap: {t_dev*1e3:.3f} ms, {len(out)} hits  ({t_scipy / t_dev:.0f}x)")
assert len(out) == len(idx)
mcf = util.to_coords_features(torch.from_numpy(map_pts[:, :3]).float().cuda(), "map", 0.1)
scf = util.to_coords_features(s_dev.float(), "scan", 0.1)
for _ in range(3): util.prune(mcf, scf, 0.1)
t = time.time()
for _ in range(20): subv, nsv = util.prune(mcf, scf, 0.1)
torch.cuda.synchronize()
print(f"device variant-B prune: {(time.time()-t)/20*1e3:.3f} ms, {len(subv)} submap voxels, {nsv} scan voxels")
