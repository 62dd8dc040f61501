"""Streaming stable-points filter: the per-scan pipeline of the reference's online ROS node
(c_ws/src/sps_filter/scripts/sps_node.py:88-176) without the ROS transport:

    pose transform (:103)  ->  variant-B submap against the device-resident map hash (:111-115)
    ->  infer (:120)  ->  epsilon filter, keep score <= eps (:148)  ->  compacted filtered cloud
    + per-stage timers T / P / I like the node's log line (:164-176).

The map voxel hash is built once (the reference re-hashes the whole map in every callback, util.py:86-89).
Everything after the pose transform stays on the device.
"""
from __future__ import annotations

import time
from dataclasses import dataclass

import numpy as np
import torch

from .datasets import util


@dataclass
class FilterResult:
    filtered: torch.Tensor     # [m, 3] scan points (sensor frame, as received) with score <= eps
    scores: torch.Tensor       # [n] stability score of every scan point
    n_scan_voxels: int         # S in the node's log line
    n_submap_voxels: int       # M
    t_total: float
    t_prune: float
    t_infer: float


class StableFilter:
    def __init__(self, model, map_points: torch.Tensor, voxel_size: float = 0.1, epsilon: float = 0.84,
                 device: str = "cuda"):
        self.model, self.ds, self.epsilon, self.device = model, float(voxel_size), float(epsilon), device
        map_xyz = torch.as_tensor(map_points)[:, :3].to(torch.float32).to(device)
        self.map_cf = util.to_coords_features(map_xyz, "map", self.ds, device)       # sps_node.py:69-74

    @torch.no_grad()
    def __call__(self, scan_xyz: np.ndarray | torch.Tensor, pose: np.ndarray | None = None) -> FilterResult:
        t0 = time.time()
        raw = torch.as_tensor(scan_xyz)[:, :3]
        world = raw
        if pose is not None:                                                           # sps_node.py:103
            world = torch.from_numpy(util.transform_point_cloud(np.asarray(raw, dtype=np.float64), np.asarray(pose)))
        world = world.to(torch.float32).to(self.device)
        t1 = time.time()
        scan_cf = util.to_coords_features(world, "scan", self.ds, self.device)         # :111
        submap, n_scan_vox = util.prune(self.map_cf, scan_cf, self.ds)                 # :115
        torch.cuda.synchronize()
        t2 = time.time()
        scores, _ = util.infer(world, submap, self.model, self.device)                 # :120
        keep = scores <= self.epsilon                                                  # :148 (<=, not <)
        filtered = torch.as_tensor(raw).to(torch.float32).to(self.device)[keep]
        torch.cuda.synchronize()
        t3 = time.time()
        return FilterResult(filtered, scores, int(n_scan_vox), int(len(submap)), t3 - t0, t2 - t1, t3 - t2)
