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

    def setup(self, stage=None, device_items=False):
        """``device_items``: the per-item work (radius submap, stacking, collate, augmentation) runs on the GPU
        (DeviceItemLoader) instead of in DataLoader workers with scipy KD-trees; same batches, same random draws."""
        if device_items:
            if self.test:
                self.test_loader = DeviceItemLoader(self.cfg, self.test_scans, self.map, shuffle=False)
            else:
                self.train_loader = DeviceItemLoader(self.cfg, self.train_scans, self.map, split='train',
                                                     shuffle=self.cfg["DATA"]["SHUFFLE"])
                self.valid_loader = DeviceItemLoader(self.cfg, self.val_scans, self.map, shuffle=False)
            return
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


class DeviceItemLoader:
    """``DataLoader(BacchusDataset(...), batch_size, shuffle, collate_fn)`` (blt_dataset.py:102-118,185-278) with every
    per-item step on the MI355X: the cached scan goes to the device raw, ``sps_radius_item`` writes the scan rows and the
    radius-submap rows of every item of the batch into one [N, 6] tensor (collate layout), the training augmentation is
    applied to each item's xyz there.  One host synchronisation per batch (the row counts).  The global torch generator is
    consumed exactly as a DataLoader with ``num_workers=0`` consumes it (base-seed draw, RandomSampler seed draw, then the
    augmentation draws item by item), so a seeded run sees the same batches as the host path."""

    def __init__(self, cfg, scans, pc_map, split=None, shuffle=False, device=None, row_factor: float = 2.5,
                 shard=(0, 1), even_shards=False):
        """``shard`` = (rank, world): this process takes every world-th batch position of the common order (every rank draws
        the same permutation from its identically seeded generator); ``even_shards`` drops the tail so that every rank takes
        the same number of steps (training: one gradient all-reduce per step)."""
        self.shard = (int(shard[0]), int(shard[1]))
        self.even_shards = bool(even_shards)
        self.cfg = cfg
        self.scans = scans
        self.map = pc_map
        self.batch_size = int(cfg["TRAIN"]["BATCH_SIZE"])
        self.shuffle = bool(shuffle)
        self.augment = bool(cfg["TRAIN"]["AUGMENTATION"]) and split == "train"
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.row_factor = float(row_factor)
        self._submap = None
        self._raw = self._pin = self._rows = self._nrows = None
        self.dataset = scans         # len(loader.dataset), as callers of a DataLoader expect

    def _batches(self, order):
        groups = [order[i: i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        rank, world = self.shard
        if self.even_shards:
            groups = groups[: len(groups) // world * world]
        return groups[rank::world]

    def __len__(self):
        return len(self._batches(list(range(len(self.scans)))))

    def _ctx(self):
        from ..models import models
        with torch.cuda.device(self.device):
            cx = models.get_context(self.device.index or 0, torch.cuda.current_stream().cuda_stream)
            if self._submap is None or self._submap.ctx is not cx:
                self._submap = DeviceRadiusSubmap(self.map[:, :3], self.cfg["MODEL"]["VOXEL_SIZE"], device=self.device, ctx=cx)
                self._calibrate_rows()
        return cx

    def _calibrate_rows(self, samples: int = 4) -> None:
        """Item rows per scan point of THIS map, from the radius query of a few sample scans (1.25 x the largest (n + m) / n,
        as ScanEngine.calibrate_rows): a map at the network's voxel size gives ~6, so the default of 2.5 made the first
        batch of every run overflow its buffer and run again.  One-off, synchronises."""
        if not len(self.scans):
            return
        step = max(1, len(self.scans) // samples)
        worst = 1.0
        for s in list(self.scans)[::step][:samples]:
            s = np.asarray(s)
            if len(s):
                _, counts = self._submap.query(s[:, :3])
                worst = max(worst, 1.0 + float(counts.sum().item()) / len(s))
        self.row_factor = max(self.row_factor, 1.25 * worst)

    def collate(self, idxs) -> torch.Tensor:
        """[sum N, 6] float32 device tensor = collate_fn([dataset[i] for i in idxs])."""
        from .._native import ERR_ITEMCAP, SpsError
        arrs = [np.asarray(self.scans[i]) for i in idxs]
        f64 = arrs[0].dtype == np.float64
        tdt, esz = (torch.float64, 8) if f64 else (torch.float32, 4)
        n_tot = int(sum(len(a) for a in arrs))
        cx = self._ctx()
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream()
            if self._pin is None or self._pin.dtype != tdt or self._pin.shape[0] < n_tot:
                st.synchronize()
                self._pin = torch.empty((n_tot + n_tot // 4 + 1024, 4), dtype=tdt).pin_memory()
                self._raw = torch.empty_like(self._pin, device=self.device)
            self._nrows = torch.zeros(max(len(arrs), 4), dtype=torch.int32, device=self.device)
            host = self._pin.numpy()
            o = 0
            for a in arrs:
                host[o: o + len(a)] = a[:, :4]
                o += len(a)
            while True:
                cap = int(n_tot * self.row_factor) + 1024
                if self._rows is None or self._rows.shape[0] < cap:
                    self._rows = torch.empty((cap, 6), dtype=torch.float32, device=self.device)
                self._raw[:n_tot].copy_(self._pin[:n_tot], non_blocking=True)
                o = 0
                for j, a in enumerate(arrs):
                    cx.radius_item(self._raw.data_ptr() + o * 4 * esz, f64, 4, len(a), float(j),
                                   None if j == 0 else self._nrows[j - 1:].data_ptr(), self._rows.data_ptr(), 6,
                                   self._rows.shape[0], self._nrows[j:].data_ptr(), st.cuda_stream)
                    o += len(a)
                ends = self._nrows[: len(arrs)].tolist()          # the one synchronisation of the batch
                try:
                    cx.check_errors(st.cuda_stream)
                    break
                except SpsError as e:
                    if e.code != ERR_ITEMCAP:
                        raise
                    self.row_factor *= 2.0                         # denser map than the buffers were sized for
            batch = self._rows[: ends[-1]].clone()
            if self.augment:                                      # blt_dataset.py:240-242: scan and submap of an item together
                a0 = 0
                for b0 in ends:
                    batch[a0:b0, 1:4] = BacchusDataset.augment_data(None, batch[a0:b0, 1:4])
                    a0 = b0
        return batch

    def __iter__(self):
        n = len(self.scans)
        torch.empty((), dtype=torch.int64).random_()              # DataLoader's per-iterator base seed (worker seeding)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))      # RandomSampler's own generator
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        mine = {tuple(g) for g in self._batches(order)}
        for i in range(0, n, self.batch_size):
            idxs = order[i: i + self.batch_size]
            if tuple(idxs) in mine:
                yield self.collate(idxs)
            elif self.augment:
                # another rank's batch: take (and drop) the augmentation draws of its items, so that the union of the
                # ranks' batches is exactly the single-process epoch and every rank's generator stays in step
                for _ in idxs:
                    torch.rand(1), torch.randn(3), torch.rand(1), torch.rand(1), torch.rand(1, 3)
