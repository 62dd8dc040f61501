"""Per-tile present-offset statistics of the kernel maps for the config-2 scene (GPU)."""
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
for which in range(6):
    n = C.c_int64()
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, which, None, C.byref(n)))
    m = torch.empty((n.value, 4), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, which, m.data_ptr(), C.byref(n)))
    mm = m.cpu().numpy().view(np.uint32)
    pc = np.unpackbits(mm.view(np.uint8), axis=1).sum(1)
    pairs = sum(ctx.map_pairs(which))
    K = 125 if which == 5 else 81
    lvl = 0 if which == 5 else which
    print(f"map {which} (K={K}, level {lvl}): tiles {n.value}, present offsets/tile mean {pc.mean():.1f} max {pc.max()}, "
          f"pairs/row {pairs / V[lvl]:.1f}, tile-MFMA efficiency {pairs / (pc.sum() * 16):.2f}")
