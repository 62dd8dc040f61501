"""Deterministic LiDAR-like synthetic scenes.

There is no dataset and no checkpoint in the build environment (reference Readme.md:104-107
are download links), so every benchmark / parity input is generated here:

  * sensor 1.8 m above a ground plane, 64 beams (elevation -22.5..+22.5 deg) x ``n_azimuth``
    azimuth steps, two walls (y=-6.0, y=+7.5, |z|<4.3), 40 vertical poles, range cut 60 m,
    Gaussian range noise sigma = 1 cm;
  * the "map" is the union of 5 such scans taken at x offsets {-4,-2,0,2,4} m;
  * the submap that accompanies a scan is the online variant-B one
    (reference src/sps/datasets/util.py:85-114: voxel-set intersection on the truncated grid,
    returned as voxel corner points).

Pure numpy, host side; used by bench.py, tests and scripts/predict.py --synthetic.
"""
from __future__ import annotations

import numpy as np

SCAN_TIMESTAMP = 1   # reference src/sps/datasets/util.py:20-21
MAP_TIMESTAMP = 0


def _scene_poles(scene_seed: int, n_poles: int = 40, extent: float = 30.0):
    rng = np.random.default_rng(10_000 + scene_seed)
    c = rng.uniform(-extent, extent, size=(n_poles, 2))
    r = rng.uniform(0.15, 0.45, size=n_poles)
    return c, r


def lidar_scan(seed: int, x_offset: float = 0.0, n_beams: int = 64, n_azimuth: int = 1750,
               scene_seed: int = 0, max_range: float = 60.0, noise: float = 0.01,
               length_scale: float = 1.0) -> np.ndarray:
    """One scan in the world frame -> float32 [n,4] = (x,y,z,label~U(0,1))."""
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(-22.5, 22.5, n_beams))
    az = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False)
    el, az = np.meshgrid(el, az, indexing="ij")
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
    o = np.array([x_offset, 0.0, 0.0])
    n = len(d)
    t = np.full(n, np.inf)

    with np.errstate(divide="ignore", invalid="ignore"):
        # ground z = -1.8
        tg = (-1.8 - o[2]) / d[:, 2]
        t = np.where((d[:, 2] < 0) & (tg > 0), np.minimum(t, tg), t)
        # walls
        for ywall in (-6.0, 7.5):
            tw = (ywall - o[1]) / d[:, 1]
            zhit = o[2] + tw * d[:, 2]
            ok = (tw > 0) & (np.abs(zhit) < 4.3)
            t = np.where(ok, np.minimum(t, tw), t)
        # poles: |o_xy + t d_xy - c|^2 = r^2
        centres, radii = _scene_poles(scene_seed, extent=30.0 * length_scale)
        dxy = d[:, :2]
        a = np.sum(dxy * dxy, axis=1)
        for c, r in zip(centres, radii):
            oc = o[:2] - c
            bq = 2.0 * (dxy @ oc)
            cq = oc @ oc - r * r
            disc = bq * bq - 4 * a * cq
            tp = (-bq - np.sqrt(np.maximum(disc, 0))) / (2 * a)
            zhit = o[2] + tp * d[:, 2]
            ok = (disc > 0) & (tp > 0) & (np.abs(zhit) < 4.3)
            t = np.where(ok, np.minimum(t, tp), t)

    keep = np.isfinite(t) & (t < max_range)
    t = t[keep] + rng.normal(0.0, noise, size=int(keep.sum()))
    pts = o[None, :] + t[:, None] * d[keep]
    label = rng.uniform(0.0, 1.0, size=len(pts))
    return np.concatenate([pts, label[:, None]], axis=1).astype(np.float32)


def build_map(offsets=(-4.0, -2.0, 0.0, 2.0, 4.0), first_seed: int = 2, **kw) -> np.ndarray:
    """Union of scans -> float32 [M,4] (x,y,z,label) in the layout of base_map.asc.npy
    (reference src/sps/datasets/blt_dataset.py:49-55)."""
    parts = [lidar_scan(first_seed + i, x_offset=x, **kw) for i, x in enumerate(offsets)]
    return np.concatenate(parts, axis=0)


def truncated_voxels(xyz: np.ndarray, ds: float) -> np.ndarray:
    """(xyz / ds).int() in float32 -- reference util.py:72-75 (truncation toward zero)."""
    q = np.array([ds, ds, ds], dtype=np.float32)
    return np.trunc(xyz[:, :3].astype(np.float32) / q).astype(np.int32)


def submap_voxel_host(map_xyz: np.ndarray, scan_xyz: np.ndarray, ds: float) -> np.ndarray:
    """Host (numpy) variant-B submap, used only to PREPARE synthetic inputs when no GPU is
    involved (CPU tests).  The product path is ``sps_amd.datasets.util.prune`` on the device.
    Rows are in scan-voxel first-occurrence order."""
    sv = truncated_voxels(scan_xyz, ds)
    mv = truncated_voxels(map_xyz, ds)

    def pack(v):
        v = v.astype(np.int64) + (1 << 20)
        return (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]

    sk = pack(sv)
    _, first = np.unique(sk, return_index=True)
    first.sort()
    su = sv[first]
    hit = np.isin(pack(su), pack(mv))
    return (su[hit].astype(np.float32) * np.float32(ds)).astype(np.float32)


def assemble(scan: np.ndarray, submap_xyz: np.ndarray, batch_index: int = 0) -> np.ndarray:
    """[N,6] = (b,x,y,z,t,label): scan rows (t=1, own label) first, then submap rows
    (t=0, label 1) -- reference blt_dataset.py:209-244 + collate_fn :173-182."""
    n, m = len(scan), len(submap_xyz)
    out = np.empty((n + m, 6), dtype=np.float32)
    out[:, 0] = batch_index
    out[:n, 1:4] = scan[:, :3]
    out[:n, 4] = SCAN_TIMESTAMP
    out[:n, 5] = scan[:, 3]
    out[n:, 1:4] = submap_xyz[:, :3]
    out[n:, 4] = MAP_TIMESTAMP
    out[n:, 5] = 1.0
    return out


def make_scene(scan_seed: int = 1, x_offset: float = 0.0, voxel_size: float = 0.1,
               n_azimuth: int = 1750, n_beams: int = 64, batch_index: int = 0,
               map_points: np.ndarray | None = None, **kw):
    """BASELINE config-2 input: returns dict(batch [N,6] float32, scan, map, n_scan)."""
    if map_points is None:
        map_points = build_map(n_azimuth=n_azimuth, n_beams=n_beams, **kw)
    scan = lidar_scan(scan_seed, x_offset=x_offset, n_azimuth=n_azimuth, n_beams=n_beams, **kw)
    sub = submap_voxel_host(map_points[:, :3], scan[:, :3], voxel_size)
    return dict(batch=assemble(scan, sub, batch_index), scan=scan, map=map_points, n_scan=len(scan))


def small_scene(seed: int = 0, n_scan: int = 2000, extent: float = 6.0, voxel_size: float = 0.1):
    """Small random surface-like cloud for fast parity tests: points on a few planes and
    a sphere, scan + jittered 'map' copy.  Returns [N,6] float32 batch."""
    rng = np.random.default_rng(seed)
    k = n_scan // 3
    p1 = np.stack([rng.uniform(-extent, extent, k), rng.uniform(-extent, extent, k),
                   np.full(k, -1.8) + rng.normal(0, 0.01, k)], 1)
    p2 = np.stack([rng.uniform(-extent, extent, k), np.full(k, 2.5) + rng.normal(0, 0.01, k),
                   rng.uniform(-1.8, 2.0, k)], 1)
    u = rng.normal(size=(n_scan - 2 * k, 3))
    p3 = 1.5 * u / np.linalg.norm(u, axis=1, keepdims=True) + np.array([-2.0, -2.0, 0.0])
    scan_xyz = np.concatenate([p1, p2, p3], 0)
    scan = np.concatenate([scan_xyz, rng.uniform(0, 1, (len(scan_xyz), 1))], 1).astype(np.float32)
    map_xyz = (scan_xyz + rng.normal(0, 0.03, scan_xyz.shape)).astype(np.float32)
    sub = submap_voxel_host(map_xyz, scan[:, :3], voxel_size)
    return assemble(scan, sub, 0)


def make_nclt_scene(seed: int = 5, n_azimuth: int = 1000, voxel_size: float = 0.1, batch_index: int = 0):
    """BASELINE config 4 (NCLT-like parking lot, SURVEY 8(d): "3x angular density or 3 merged scans -> ~300k active
    voxels at tensor stride 1, map 5x larger"): three merged 128-beam scans taken 15 m apart, range 100 m, against a
    map of 25 scan positions (5x the 5 of config 2).  ~500k rows, >= 300k active level-0 voxels."""
    kw = dict(n_azimuth=n_azimuth, n_beams=128, max_range=100.0)
    map_points = build_map(offsets=tuple(np.linspace(-24.0, 24.0, 25)), **kw)
    scan = np.concatenate([lidar_scan(10 * seed + i, x_offset=x, **kw) for i, x in enumerate((-15.0, 0.0, 15.0))], 0)
    sub = submap_voxel_host(map_points[:, :3], scan[:, :3], voxel_size)
    return dict(batch=assemble(scan, sub, batch_index), scan=scan, map=map_points, n_scan=len(scan))


def sequence_map(n_scans: int, step: float = 0.5, spacing: float = 2.0, **kw) -> np.ndarray:
    """Map for a config-3 sequence: scans every ``spacing`` m along the sensor path (x = 0 .. n_scans * step)."""
    xs = np.arange(-4.0, n_scans * step + 4.0 + 1e-9, spacing)
    return build_map(offsets=tuple(xs), **kw)


def make_sequence(n_scans: int, step: float = 0.5, first_seed: int = 100, voxel_size: float = 0.1,
                  map_points: np.ndarray | None = None, **kw):
    """BASELINE config 3: consecutive scans, the sensor advancing ``step`` m per scan along x; yields [N,6] batches
    (b = 0) in the layout of BacchusDataset.__getitem__ + collate_fn."""
    if map_points is None:
        map_points = sequence_map(n_scans, step, **kw)
    for i in range(n_scans):
        scan = lidar_scan(first_seed + i, x_offset=step * i, **kw)
        sub = submap_voxel_host(map_points[:, :3], scan[:, :3], voxel_size)
        yield assemble(scan, sub, 0)


def collate(items, ) -> np.ndarray:
    """BacchusModule.collate_fn (blt_dataset.py:173-182) on numpy [n_i, 6] batches: batch column = position in the list."""
    out = []
    for b, it in enumerate(items):
        q = it.copy()
        q[:, 0] = b
        out.append(q)
    return np.concatenate(out, 0)
