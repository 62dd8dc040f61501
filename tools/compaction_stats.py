"""How many 16-row MFMA slots would row compaction over R-row supertiles need, vs the current per-tile scheme?"""
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
for lvl in range(5):
    n = C.c_int64()
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, lvl, None, C.byref(n)))
    m = torch.empty((n.value, 4), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, lvl, m.data_ptr(), C.byref(n)))
    bits = np.unpackbits(m.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :81].astype(bool)   # [tiles,81]
    nb = torch.empty((81, V[lvl]), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_nbr(ctx.handle, lvl, nb.data_ptr()))
    pres = (nb.cpu().numpy() >= 0) & np.repeat(bits, 16, axis=0)[: V[lvl]].T          # [81, V] valid presence
    cur = bits.sum()
    line = f"level {lvl}: V {V[lvl]}, pairs {pres.sum()}, current tile-slots {cur} (eff {pres.sum() / (cur * 16):.2f})"
    for R in (32, 64, 128):
        pad = (-V[lvl]) % R
        p = np.pad(pres, ((0, 0), (0, pad))).reshape(81, -1, R).sum(2)                   # rows with neighbour k per supertile
        slots = np.ceil(p / 16).sum()
        line += f" | R={R}: slots {int(slots)} ({cur / slots:.2f}x fewer, eff {pres.sum() / (slots * 16):.2f})"
    print(line)
