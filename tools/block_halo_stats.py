"""CPU estimate for a block-halo convolution: rows per 4x4x4 block, present adjacent blocks, halo rows per block,
for every level of the config-2 scene (numpy only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import sps_oracle as O
from sps_amd import synthetic
b = synthetic.make_scene(scan_seed=1)["batch"]
q = O.quantize(b[:, :5], 0.1)
vox, _ = O.unique_first(q)
for lvl in range(5):
    ts = 1 << lvl
    if lvl:
        vox = np.unique(np.concatenate([vox[:, :1], (vox[:, 1:4] // ts) * ts, vox[:, 4:5]], 1), axis=0)
    c = vox.copy()
    c[:, 1:4] //= ts                       # level units
    blk = np.concatenate([c[:, :1], c[:, 1:4] >> 2, c[:, 4:5]], 1)
    ub, inv, cnt = np.unique(blk, axis=0, return_inverse=True, return_counts=True)
    key = {tuple(r): i for i, r in enumerate(ub)}
    nadj, halo = [], []
    for i, r in enumerate(ub):
        n = 0; h = 0
        for dt in (-1, 0, 1):
            for dz in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        j = key.get((r[0], r[1] + dx, r[2] + dy, r[3] + dz, r[4] + dt))
                        if j is not None:
                            n += 1; h += cnt[j]
        nadj.append(n); halo.append(h)
    nadj, halo = np.array(nadj), np.array(halo)
    tiles_blockaligned = np.ceil(cnt / 16).sum()
    print(f"level {lvl}: rows {len(vox)}, blocks {len(ub)}, rows/block mean {cnt.mean():.1f} p90 {np.percentile(cnt,90):.0f} max {cnt.max()}, "
          f"adjacent blocks present mean {nadj.mean():.1f}, halo rows/block mean {halo.mean():.0f} p90 {np.percentile(halo,90):.0f} max {halo.max()}, "
          f"halo rows total / rows {halo.sum()/len(vox):.1f}, 16-row tiles: packed {int(np.ceil(len(vox)/16))} block-aligned {int(tiles_blockaligned)}")
