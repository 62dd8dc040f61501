"""Time the baseline heads at their deployed sizes on the GPU:
  4DMOS : buffer of 10 consecutive config-2 scans (t = scan index), voxel 0.2 (mos4d_node.py:57,98-110)
  MapMOS: one config-2 scan (t = 0, index 1) + the map points within 30 m (t = -1, index 0), voxel 0.1
          (mapmos_node.py:64-95)
Usage: python tools/heads_timing.py [--iters 50]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sps_amd import synthetic                                   # noqa: E402
from sps_amd.models.baselines import MapMOSNet, MOS4DNet         # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    torch.manual_seed(0)
    out = {}
    # ---- 4DMOS
    rows = []
    for i in range(10):
        s = synthetic.lidar_scan(seed=40 + i, x_offset=0.5 * i)
        rows.append(np.concatenate([np.zeros((len(s), 1), np.float32), s[:, :3], np.full((len(s), 1), 500 + i, np.float32)], 1))
    c = torch.from_numpy(np.concatenate(rows, 0).astype(np.float32)).cuda()
    m4 = MOS4DNet(0.2).cuda().eval().freeze()
    # the t shift is known to a node (its own scan counter); time the forward without the host read of min(t)
    from sps_amd.models import baselines
    tb = baselines._t_base(c)
    baselines_t = baselines._t_base
    baselines._t_base = lambda coords: tb
    ms = timed(lambda: m4(c), args.iters)
    baselines._t_base = baselines_t
    from sps_amd.models.models import get_context
    out["mos4d"] = {"points": int(c.shape[0]), "voxels": get_context(0).level_counts(), "ms": round(ms, 3)}
    # ---- MapMOS
    scan = synthetic.lidar_scan(seed=1)[:, :3]
    mp = synthetic.build_map()[:, :3]
    mp = mp[np.sqrt((mp ** 2).sum(1)) <= 30.0]
    st, mt = torch.from_numpy(scan).cuda(), torch.from_numpy(mp).cuda()
    si, mi = torch.ones(len(scan), 1, device="cuda"), torch.zeros(len(mp), 1, device="cuda")
    mm = MapMOSNet(0.1).cuda().eval().freeze()
    ms = timed(lambda: mm.predict(st, mt, si, mi), args.iters)
    out["mapmos"] = {"scan_points": len(scan), "map_points": len(mp), "voxels": get_context(0).level_counts(),
                     "ms": round(ms, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
