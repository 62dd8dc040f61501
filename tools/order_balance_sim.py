"""How evenly do the dealing rules put a launch's work on the CUs?  (CPU only; the config-2 scene.)
A launch's workgroups are all resident and land on CU (linear id mod 256); positions p, p + ways, p + 2 ways ... share a CU
(ways = 256 / column groups).  Items (16-row tiles weighted by present offsets at levels 2-4; 64-row supertiles weighted by
chunk count at levels 0-1) are sorted heaviest first; a rule maps position -> sorted index.  Prints max / mean load per way:
  r5     rounds 3-5: boustrophedon over tiers, the last partial tier as it comes
  split  the ways that hold one item more take all their items from the light end (the product rule for q = n / ways <= 2)
  tail   boustrophedon with the tiers paired from the END (the last full tier runs backwards), partial tier forwards
  lpt    greedy longest-processing-time with the position structure's item counts per way (reference: not parallel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sps_amd import synthetic
from oracle import sps_oracle as O

pack = lambda a: (((a[:, 0].astype(np.int64) * 32 + (a[:, 4] + 16)) * (1 << 18) + (a[:, 3] + (1 << 17))) * (1 << 18)
                  + (a[:, 2] + (1 << 17))) * (1 << 18) + (a[:, 1] + (1 << 17))


def presence(co0, l):
    co = co0.copy()
    co[:, 1:4] >>= l
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
        pres[:, k] = ks[np.minimum(np.searchsorted(ks, q), V - 1)] == q
    return pres


def r5(pos, n, W):
    tier, way = divmod(pos, W)
    ln = min(W, n - tier * W)
    return tier * W + ((ln - 1 - way) if tier & 1 else way)


def split(pos, n, W):
    tier, way = divmod(pos, W)
    q, r = divmod(n, W)
    if way < r:
        return q * (W - r) + tier * r + ((r - 1 - way) if tier & 1 else way)
    wh, c = W - r, way - r
    return tier * wh + ((wh - 1 - c) if tier & 1 else c)


def tail(pos, n, W):
    tier, way = divmod(pos, W)
    q = n // W
    if tier >= q:
        return tier * W + way
    return tier * W + ((W - 1 - way) if (q - 1 - tier) % 2 == 0 else way)


def evaluate(w, W, name):
    w = np.sort(np.asarray(w, float))[::-1]
    n = len(w)
    out = []
    for rule in (r5, split, tail):
        s = np.zeros(W)
        idx = [rule(p, n, W) for p in range(n)]
        assert sorted(idx) == list(range(n))
        for p, i in enumerate(idx):
            s[p % W] += w[i]
        out.append(f"{rule.__name__} {s.max() / (w.sum() / W):.3f}")
    q, r = divmod(n, W)
    s, cnt = np.zeros(W), np.zeros(W, int)
    capw = np.where(np.arange(W) < r, q + 1, q)
    for x in w:                                     # greedy LPT under the per-way item counts
        ok = cnt < capw
        j = np.flatnonzero(ok)[np.argmin(s[ok])]
        s[j] += x; cnt[j] += 1
    out.append(f"lpt {s.max() / (w.sum() / W):.3f}")
    print(f"{name:44s} n = {n:5d} ways = {W:3d} (q = {q}, r = {r:3d})  max / mean load per way: " + "  ".join(out))


co0 = O.quantize(synthetic.make_scene(scan_seed=1)["batch"][:, :5], 0.1)
for l in (0, 1):
    pres = presence(co0, l)
    V = len(pres)
    ch = np.ceil(np.pad(pres, ((0, (-V) % 64), (0, 0))).reshape(-1, 64, 81).sum(1) / 16).sum(1)
    evaluate(ch, 256, f"level {l}: supertiles by chunk count (k_conv_px)")
for l in (2, 3, 4):
    pres = presence(co0, l)
    V = len(pres)
    t = np.pad(pres, ((0, (-V) % 16), (0, 0))).reshape(-1, 16, 81).any(1).sum(1)
    evaluate(t, 256, f"level {l}: tiles by present offsets, 1 column group")
    if l >= 3:
        evaluate(t, 128, f"level {l}: tiles by present offsets, 2 column groups")


def tiersplit(pos, n, W):
    """every FULL tier keeps its own items (dispatch stays heaviest-first tier by tier); inside a tier the r ways that will also
    hold an item of the partial tier take the tier's r LIGHTEST items, the other ways its W - r heaviest; alternating directions"""
    tier, way = divmod(pos, W)
    q, r = divmod(n, W)
    if tier >= q:
        return tier * W + way
    if way < r:
        c = (r - 1 - way) if tier & 1 else way
        return tier * W + (W - r) + c
    wh, c = W - r, way - r
    return tier * W + ((wh - 1 - c) if tier & 1 else c)


if __name__ == "__main__":
    print("\nwith the tier-wise split (dispatch order stays heaviest-first by tier):")
    for l in (0, 1):
        pres = presence(co0, l)
        V = len(pres)
        ch = np.sort(np.ceil(np.pad(pres, ((0, (-V) % 64), (0, 0))).reshape(-1, 64, 81).sum(1) / 16).sum(1))[::-1]
        for rule in (r5, split, tiersplit):
            s = np.zeros(256)
            for p in range(len(ch)):
                s[p % 256] += ch[rule(p, len(ch), 256)]
            print(f"level {l} px: {rule.__name__:10s} {s.max() / (ch.sum() / 256):.3f}")
    for l, W in ((2, 256), (3, 256), (3, 128)):
        pres = presence(co0, l)
        V = len(pres)
        t = np.sort(np.pad(pres, ((0, (-V) % 16), (0, 0))).reshape(-1, 16, 81).any(1).sum(1))[::-1].astype(float)
        for rule in (r5, split, tiersplit):
            s = np.zeros(W)
            idx = [rule(p, len(t), W) for p in range(len(t))]
            assert sorted(idx) == list(range(len(t)))
            for p, i in enumerate(idx):
                s[p % W] += t[i]
            print(f"level {l} tiles, {W} ways: {rule.__name__:10s} {s.max() / (t.sum() / W):.3f}")
