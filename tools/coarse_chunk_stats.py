"""VERDICT r5 item 1b, priced before it is built: how many MFMA slots a PAIR-EXACT kernel would execute at levels 2-4
(16-pair chunks of one offset, compacted inside supertiles of S output rows) against what k_conv executes now (every
offset any row of a 16-row tile has, for all 16 rows).  CPU only: numpy re-creation of the block-contiguous row order of
every level of the config-2 scene (as tools/granularity_stats.py does for level 0).
Prints per level: rows, pairs, k_conv's (tile, offset) slots, and per S the chunk count, slots per pair and the MFMA count
of the level's widest layer (chunks x C_in / 4 x C_out / 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sps_amd import synthetic
from oracle import sps_oracle as O

LAYERS = {2: ("block6.conv1", 48, 32), 3: ("block5.conv1", 96, 64), 4: ("block4.conv2", 64, 64), 1: ("block7.conv1", 24, 16), 0: ("block8.conv1", 16, 8)}
co0 = O.quantize(synthetic.make_scene(scan_seed=1)["batch"][:, :5], 0.1)          # [N,5] = b, x, y, z, t
pack = lambda a: (((a[:, 0].astype(np.int64) * 32 + (a[:, 4] + 16)) * (1 << 18) + (a[:, 3] + (1 << 17))) * (1 << 18)
                  + (a[:, 2] + (1 << 17))) * (1 << 18) + (a[:, 1] + (1 << 17))
for l in (2, 3, 4):
    co = co0.copy()
    co[:, 1:4] >>= l                                                               # coordinates in units of the level's stride
    u, first = np.unique(co, axis=0, return_index=True)
    blk = np.stack([u[:, 0], u[:, 4], u[:, 3] >> 2, u[:, 2] >> 2, u[:, 1] >> 2], 1)
    bu, binv = np.unique(blk, axis=0, return_inverse=True)
    binv = binv.ravel()
    bfirst = np.full(len(bu), 1 << 62, np.int64)
    np.minimum.at(bfirst, binv, first)
    bit = ((u[:, 3] & 3) << 4) | ((u[:, 2] & 3) << 2) | (u[:, 1] & 3)
    c = u[np.lexsort((bit, bfirst[binv]))]
    V = len(c)
    ks = np.sort(pack(c))
    pres = np.zeros((V, 81), bool)
    for k in range(81):
        n = c.copy()
        n[:, 1] += k % 3 - 1; n[:, 2] += k // 3 % 3 - 1; n[:, 3] += k // 9 % 3 - 1; n[:, 4] += k // 27 - 1
        q = pack(n)
        pos = np.minimum(np.searchsorted(ks, q), V - 1)
        pres[:, k] = ks[pos] == q
    P = int(pres.sum())
    name, cin, cout = LAYERS[l]
    per_chunk = (cin // 4) * (cout // 16)
    t16 = np.pad(pres, ((0, (-V) % 16), (0, 0))).reshape(-1, 16, 81).any(1)
    slots16 = int(t16.sum())
    print(f"level {l}: rows {V}, pairs {P} ({P / V:.2f} per row), exact 16-pair MFMA groups {P / 16:.0f} -> {name} ({cin}->{cout}): "
          f"{P / 16 * per_chunk / 1e3:.0f} k MFMAs pair-exact without padding")
    print(f"  k_conv now: {slots16} (tile, offset) slots = {slots16 * 16 / P:.2f} slots per pair -> {slots16 * per_chunk / 1e3:.0f} k MFMAs")
    for S in (16, 32, 64, 128, 256):
        n = np.pad(pres, ((0, (-V) % S), (0, 0))).reshape(-1, S, 81).sum(1)
        ch = np.ceil(n / 16)
        print(f"  pair-exact, supertiles of {S:3d} rows ({len(n):4d} workgroups): {int(ch.sum()):6d} chunks = {ch.sum() * 16 / P:.2f} slots per pair "
              f"-> {ch.sum() * per_chunk / 1e3:.0f} k MFMAs ({ch.sum() / slots16:.2f} x k_conv); chunks per supertile p50 / max {np.percentile(ch.sum(1), 50):.0f} / {ch.sum(1).max():.0f}")
