"""Live fraction of (row group, offset) slots of the level-0 3x3x3x3 map at row-group sizes 4..64, and the fill of
16-pair chunks compacted per offset inside S-row supertiles (CPU only: numpy re-creation of the block-contiguous row order).
Config-2 scene: 1 884 588 pairs on 108 390 rows; live fraction 0.57 / 0.48 / 0.43 / 0.40 / 0.38 at 4 / 8 / 16 / 32 / 64 rows;
chunk fill 0.76 / 0.86 / 0.92 at S = 64 / 128 / 256 (154 511 / 137 695 / 127 949 chunks against 273 k (tile, offset) slots)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sps_amd import synthetic
from oracle import sps_oracle as O

co = O.quantize(synthetic.make_scene(scan_seed=1)["batch"][:, :5], 0.1)          # [N,5] = b, x, y, z, t
u, first = np.unique(co, axis=0, return_index=True)
blk = np.stack([u[:, 0], u[:, 4], u[:, 3] >> 2, u[:, 2] >> 2, u[:, 1] >> 2], 1)  # 4x4x4 blocks per (b, t)
bu, binv = np.unique(blk, axis=0, return_inverse=True)
binv = binv.ravel()
bfirst = np.full(len(bu), 1 << 62, np.int64)
np.minimum.at(bfirst, binv, first)                                                 # blocks in first-occurrence order
bit = ((u[:, 3] & 3) << 4) | ((u[:, 2] & 3) << 2) | (u[:, 1] & 3)
c = u[np.lexsort((bit, bfirst[binv]))]                                             # rows block-contiguous, bit order inside
V = len(c)
pack = lambda a: (((a[:, 0].astype(np.int64) * 32 + (a[:, 4] + 16)) * (1 << 18) + (a[:, 3] + (1 << 17))) * (1 << 18)
                  + (a[:, 2] + (1 << 17))) * (1 << 18) + (a[:, 1] + (1 << 17))
ks = np.sort(pack(c))
pres = np.zeros((V, 81), bool)
for k in range(81):
    n = c.copy()
    n[:, 1] += k % 3 - 1; n[:, 2] += k // 3 % 3 - 1; n[:, 3] += k // 9 % 3 - 1; n[:, 4] += k // 27 - 1
    q = pack(n)
    pos = np.minimum(np.searchsorted(ks, q), V - 1)
    pres[:, k] = ks[pos] == q
P = int(pres.sum())
print(f"rows {V}, blocks {len(bu)}, pairs {P} ({P / V:.2f} per row)")
for g in (4, 8, 16, 32, 64):
    t = np.pad(pres, ((0, (-V) % g), (0, 0))).reshape(-1, g, 81).any(1)
    print(f"row groups of {g:2d}: {t.sum(1).mean():5.1f} present offsets per group, live fraction {P / (t.sum() * g):.3f}")
for S in (64, 128, 256):
    n = np.pad(pres, ((0, (-V) % S), (0, 0))).reshape(-1, S, 81).sum(1)
    ch = np.ceil(n / 16)
    print(f"supertiles of {S:3d} rows: {int(ch.sum())} chunks of 16 pairs, fill {P / (ch.sum() * 16):.3f}, "
          f"chunks per supertile p10 / p50 / p90 = {np.percentile(ch.sum(1), 10):.0f} / {np.percentile(ch.sum(1), 50):.0f} / {np.percentile(ch.sum(1), 90):.0f}")
