"""Robustness sweep (GPU): tiny and awkward clouds (1..1000 points on a line, a cube, one voxel, random; two time slices) through the
product forward against the numpy oracle: row counts around the 16 / 64-row tile and supertile boundaries."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from oracle import sps_oracle as O
from sps_amd import synthetic
params = O.random_params(seed=3)
net = H.net_from_params(params).cuda().eval().freeze()
rng = np.random.default_rng(5)
worst = 0.0
for n in (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 257, 1000):
    for kind in ("line", "cube", "same", "rand"):
        if kind == "line":
            xyz = np.stack([np.arange(n) * 0.1 - 3.0, np.zeros(n), np.zeros(n)], 1)
        elif kind == "cube":
            k = int(np.ceil(n ** (1 / 3)))
            g = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
            xyz = g * 0.1 - 0.35
        elif kind == "same":
            xyz = np.zeros((n, 3)) + 0.05
        else:
            xyz = rng.uniform(-1.5, 1.5, (n, 3))
        t = (np.arange(n) % 2).astype(np.float32)
        batch = np.zeros((n, 6), np.float32)
        batch[:, 1:4] = xyz; batch[:, 4] = t; batch[:, 5] = rng.uniform(0, 1, n)
        ref = O.sps_forward(params, batch[:, :5], 0.1)
        ref = ref[0] if isinstance(ref, tuple) else ref
        out = net(torch.from_numpy(batch).cuda()).cpu().numpy()
        err = float(np.max(np.abs(out - np.asarray(ref).reshape(-1))))
        worst = max(worst, err)
        if err > 1e-4:
            print("MISMATCH", n, kind, err)
print("edge cases done, worst |score - oracle| =", worst)
