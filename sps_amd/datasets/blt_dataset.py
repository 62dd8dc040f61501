"""Host-side mirror of the reference's offline data assembly ``sps.datasets.blt_dataset``
(src/sps/datasets/blt_dataset.py): the $DATA tree reader, the per-scan item
[scan(x,y,z,t=1,label); submap(x,y,z,t=0,label=1)] with the KD-tree radius submap
("variant A", :224-226,258-271) and the collate layout [N,6]=(b,x,y,z,t,label) (:173-182).

pytorch_lightning is not a dependency: BacchusModule is a plain class with the same methods.
Training-time augmentation (:241-242,273-278) lives in ``augmentation.py`` and is applied to the training split only.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from scipy.spatial import cKDTree
from torch.utils.data import DataLoader, Dataset

from . import augmentation, util


class BacchusModule:
    def __init__(self, cfg, test=False):
        self.cfg = cfg
        self.test = test
        self.root_dir = str(os.environ.get("DATA"))
        split = self.cfg['DATA']['SPLIT']
        if self.test:
            print('Loading testing data ...')
            self.test_scans = self.cash_scans(*self.get_scans_poses(split['TEST']))
        else:
            print('Loading training data ...')
            self.train_scans = self.cash_scans(*self.get_scans_poses(split['TRAIN']))
            print('Loading validating data ...')
            self.val_scans = self.cash_scans(*self.get_scans_poses(split['VAL']))
        # map rows: [x, y, z, label] (:49-55)
        map_pth = os.path.join(self.root_dir, "maps", self.cfg["TRAIN"]["MAP"])
        ext = os.path.splitext(map_pth)[1]
        self.map = (np.load(map_pth) if ext == '.npy' else np.loadtxt(map_pth))[:, :4]

    def cash_scans(self, scans_pth, poses_pth, map_tr_pths):
        """:57-75 -- scan xyz <- T_map_transform . (T_pose . xyz), stored back in the file dtype."""
        cached = []
        for scan_pth, pose_pth, map_tr_pth in zip(scans_pth, poses_pth, map_tr_pths):
            scan = np.load(scan_pth)
            pose = np.loadtxt(pose_pth, delimiter=',')
            map_transform = np.loadtxt(map_tr_pth, delimiter=',')
            scan[:, :3] = util.transform_point_cloud(scan[:, :3], pose)
            scan[:, :3] = util.transform_point_cloud(scan[:, :3], map_transform)
            cached.append(scan)
        return cached

    def get_scans_poses(self, seqs):
        """:78-100 -- sorted file lists of every sequence + the per-sequence map_transform path."""
        scans, poses, transforms = [], [], []
        for sequence in seqs:
            base = os.path.join(self.root_dir, "sequence", sequence)
            s = sorted(os.path.join(base, "scans", f) for f in os.listdir(os.path.join(base, "scans")))
            p = sorted(os.path.join(base, "poses", f) for f in os.listdir(os.path.join(base, "poses")))
            scans += s
            poses += p
            transforms += [os.path.join(base, "map_transform")] * len(s)
        assert len(scans) == len(poses) == len(transforms), 'The length of those arrays should be the same!'
        return scans, poses, transforms

    def _loader(self, dataset, shuffle):
        return DataLoader(dataset=dataset, batch_size=self.cfg["TRAIN"]["BATCH_SIZE"], collate_fn=self.collate_fn,
                          shuffle=shuffle, num_workers=self.cfg["DATA"]["NUM_WORKER"], pin_memory=True,
                          drop_last=False, timeout=0)

    def setup(self, stage=None):
        if self.test:
            self.test_loader = self._loader(BacchusDataset(self.cfg, self.test_scans, self.map), False)
        else:
            self.train_loader = self._loader(BacchusDataset(self.cfg, self.train_scans, self.map, split='train'),
                                             self.cfg["DATA"]["SHUFFLE"])
            self.valid_loader = self._loader(BacchusDataset(self.cfg, self.val_scans, self.map), False)

    def train_dataloader(self):
        return self.train_loader

    def val_dataloader(self):
        return self.valid_loader

    def test_dataloader(self):
        return self.test_loader

    @staticmethod
    def collate_fn(batch):
        """:173-182 -- prepend the batch index column, stack the items: [sum N, 6]."""
        if len(batch) == 0:
            return None
        parts = []
        for i, item in enumerate(batch):
            col = torch.full((len(item), 1), i, dtype=item.dtype)
            parts.append(torch.cat([col, item], dim=1))
        return torch.cat(parts, dim=0)


class BacchusDataset(Dataset):
    """Per-scan item for point cloud prediction (:185-271)."""

    def __init__(self, cfg, scans, pc_map, split=None):
        self.cfg = cfg
        self.scans = scans
        self.dataset_size = len(scans)
        self.map = pc_map
        self.kd_tree_target = cKDTree(self.map[:, :3])          # built once (:198)
        self.augment = bool(self.cfg["TRAIN"]["AUGMENTATION"]) and split == "train"          # :200-204

    def __len__(self):
        return self.dataset_size

    def __getitem__(self, idx):
        scan = self.scans[idx]
        xyz = torch.tensor(scan[:, :3]).to(torch.float32).reshape(-1, 3)
        labels = torch.tensor(scan[:, 3]).to(torch.float32).reshape(-1, 1)
        scan_rows = torch.hstack([self.add_timestamp(xyz, util.SCAN_TIMESTAMP), labels])
        # variant-A submap: map points within VOXEL_SIZE (metric radius) of any scan point, one hit
        # list per scan point, duplicates kept (:222-226)
        submap_idx = self.select_closest_points(cKDTree(scan[:, :3]), self.kd_tree_target)
        sub_xyz = torch.tensor(self.map[submap_idx, :3]).to(torch.float32).reshape(-1, 3)
        sub_rows = torch.hstack([self.add_timestamp(sub_xyz, util.MAP_TIMESTAMP), torch.ones(sub_xyz.shape[0], 1)])
        item = torch.vstack([scan_rows, sub_rows])
        if self.augment:                                             # :240-242, training split only
            item[:, :3] = self.augment_data(item[:, :3])
        return item

    def augment_data(self, scan_map_batch):
        """:273-278 -- yaw, small rotation, mirror, scale of scan and submap together."""
        for step in (augmentation.rotate_point_cloud, augmentation.rotate_perturbation_point_cloud,
                     augmentation.random_flip_point_cloud, augmentation.random_scale_point_cloud):
            scan_map_batch = step(scan_map_batch)
        return scan_map_batch

    def add_timestamp(self, data, stamp):
        return torch.hstack([data, torch.full((len(data), 1), stamp, dtype=data.dtype)])

    def select_closest_points(self, kd_tree_ref, kd_tree_target):
        """:258-271 -- scipy query_ball_tree, hit lists concatenated in scan-point order."""
        hits = kd_tree_ref.query_ball_tree(kd_tree_target, self.cfg["MODEL"]["VOXEL_SIZE"])
        if len(hits) == 0:
            return np.zeros(0, dtype=int)
        return np.concatenate([np.asarray(h, dtype=np.int64) for h in hits]).astype(int)


class DeviceRadiusSubmap:
    """select_closest_points (:258-271) on the MI355X: a uniform grid over the map (cell size just above
    r), exact float64 radius test over the 27 neighbour cells (C ABI: sps_radius_*).  Returns, like the
    reference, the concatenation over scan points of the indices of the map points within r, duplicates
    kept; inside one scan point's list the hits are ordered by neighbour cell, ascending map index inside a
    cell (scipy's order there is the KD-tree's traversal order, i.e. unspecified)."""

    def __init__(self, map_xyz, radius: float, device="cuda", ctx=None):
        from ..models import models
        self.radius = float(radius)
        self.device = torch.device(device)
        xyz = torch.as_tensor(np.ascontiguousarray(np.asarray(map_xyz)[:, :3], dtype=np.float64)).to(self.device)
        cell = self.radius * (1.0 + 1e-7)                   # >= r, so every hit lies in the 27 cells around the query
        c = torch.floor(xyz / cell).to(torch.int64)
        assert int(c.abs().max()) < (1 << 20) - 1, "map extent exceeds the radius grid"
        keys = ((c[:, 2] + (1 << 20)) << 42) | ((c[:, 1] + (1 << 20)) << 21) | (c[:, 0] + (1 << 20))
        skeys, order = torch.sort(keys, stable=True)        # grouped by cell, ascending map index inside a cell
        ukeys, counts = torch.unique_consecutive(skeys, return_counts=True)
        start = torch.zeros(len(ukeys) + 1, dtype=torch.int32, device=self.device)
        start[1:] = torch.cumsum(counts, 0).to(torch.int32)
        pts = order.to(torch.int32).contiguous()
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream().cuda_stream
            # the context that owns the grid's device copies (others attach to it: Context.radius_grid_attach)
            self.ctx = ctx if ctx is not None else models.get_context(self.device.index or 0, self.stream)
            from .. import _native
            _native.check(_native.lib.sps_radius_grid_upload(
                self.ctx.handle, ukeys.contiguous().data_ptr(), start.data_ptr(), pts.data_ptr(), xyz.data_ptr(),
                len(ukeys), len(xyz), cell, self.radius, self.stream))

    def query(self, scan_xyz):
        """scan_xyz [n, >=3] -> (indices int64 [total], counts int32 [n]) as device tensors."""
        from .. import _native
        s = torch.as_tensor(np.ascontiguousarray(np.asarray(scan_xyz)[:, :3], dtype=np.float64)
                            if not torch.is_tensor(scan_xyz) else scan_xyz[:, :3].to(torch.float64).contiguous())
        s = s.to(self.device)
        n = len(s)
        counts27 = torch.empty(n * 27, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            _native.check(_native.lib.sps_radius_count(self.ctx.handle, s.data_ptr(), 3, n, counts27.data_ptr(), st))
            offsets = torch.zeros(n * 27 + 1, dtype=torch.int64, device=self.device)
            offsets[1:] = torch.cumsum(counts27, 0)
            total = int(offsets[-1].item())
            out = torch.empty(total, dtype=torch.int64, device=self.device)
            _native.check(_native.lib.sps_radius_fill(self.ctx.handle, s.data_ptr(), 3, n, offsets.data_ptr(),
                                                      out.data_ptr(), st))
        return out, counts27.view(n, 27).sum(1).to(torch.int32)


def device_item(dataset: BacchusDataset, idx: int, submap: DeviceRadiusSubmap) -> torch.Tensor:
    """BacchusDataset.__getitem__ (:209-244) with the submap selected on the device: float32 [n, 5] on the GPU
    = [scan (x,y,z,t=1,label); submap (x,y,z,t=0,label=1)]."""
    scan = dataset.scans[idx]
    dev = submap.device
    sub_idx, _ = submap.query(scan[:, :3])
    s = torch.as_tensor(np.asarray(scan[:, :4])).to(dev)
    scan_rows = torch.cat([s[:, :3].to(torch.float32), torch.ones(len(s), 1, device=dev), s[:, 3:4].to(torch.float32)], 1)
    m = torch.as_tensor(np.asarray(dataset.map[:, :3])).to(dev)[sub_idx].to(torch.float32)
    sub_rows = torch.cat([m, torch.zeros(len(m), 1, device=dev), torch.ones(len(m), 1, device=dev)], 1)
    return torch.cat([scan_rows, sub_rows], 0)
