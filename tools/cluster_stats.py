"""Potential of re-ordering rows inside W-row windows so that 16-row tiles hold rows with similar neighbour
masks: executed (tile, offset) slots under (a) the current order, (b) rows sorted by mask inside each window,
(c) a greedy clustering.  Level 0 and 1 maps of the config-2 scene."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sps_amd import synthetic, _native
from sps_amd.models.models import SPSNet, get_context
import bench
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
b = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
net(b); torch.cuda.synchronize()
ctx = get_context(0)
V = ctx.level_counts()
def slots(pres):            # pres [V,81] bool in execution order -> number of (16-row tile, offset) slots
    pad = (-len(pres)) % 16
    p = np.pad(pres, ((0, pad), (0, 0))).reshape(-1, 16, 81).any(1)
    return int(p.sum())
for lvl in (0, 1, 2):
    n = C.c_int64()
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, lvl, None, C.byref(n)))
    m = torch.empty((n.value, 4), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, lvl, m.data_ptr(), C.byref(n)))
    bits = np.unpackbits(m.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :81].astype(bool)
    nb = torch.empty((81, V[lvl]), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_nbr(ctx.handle, lvl, nb.data_ptr()))
    pres = ((nb.cpu().numpy() >= 0) & np.repeat(bits, 16, axis=0)[: V[lvl]].T).T.copy()      # [V,81]
    base = slots(pres)
    line = f"level {lvl}: pairs {pres.sum()}, current slots {base} (eff {pres.sum() / (base * 16):.2f})"
    for W in (64, 256, 1024):
        order = np.arange(len(pres))
        key = np.packbits(pres, axis=1)                                   # 11 bytes per row
        for w0 in range(0, len(pres), W):
            seg = slice(w0, min(w0 + W, len(pres)))
            k = key[seg]
            idx = np.lexsort(k.T[::-1])                                   # sort rows of the window by mask
            order[seg] = np.arange(seg.start, seg.stop)[idx]
        s = slots(pres[order])
        line += f" | sort W={W}: {s} ({base / s:.2f}x)"
    print(line, flush=True)
