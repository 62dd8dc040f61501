"""Work-balance statistics of the 16-row conv tiles of one level (config-2 scene): present offsets per tile, spread
inside a 4-tile workgroup, and what a perfect balance inside the workgroup / across the launch would buy."""
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
for level in (0, 1):
    n = C.c_int64()
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, level, None, C.byref(n)))
    m = torch.empty((n.value, 4), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, level, m.data_ptr(), C.byref(n)))
    mm = m.cpu().numpy().view(np.uint32)[:, :3]
    nk = np.zeros(len(mm), np.int64)
    for w in range(3):
        nk += np.array([bin(int(x)).count("1") for x in mm[:, w]])
    pad = (-len(nk)) % 4
    g = np.concatenate([nk, np.zeros(pad, np.int64)]).reshape(-1, 4)
    print(f"level {level}: {len(nk)} tiles, offsets per tile mean {nk.mean():.1f} p10 {np.percentile(nk,10):.0f} p50 {np.percentile(nk,50):.0f} "
          f"p90 {np.percentile(nk,90):.0f} max {nk.max()}")
    print(f"   max / mean over the launch {nk.max() / nk.mean():.2f}; inside a 4-tile workgroup: mean of (max / mean) "
          f"{np.mean(g.max(1) / np.maximum(g.mean(1), 1e-9)):.2f}; workgroup sums: p90 / mean {np.percentile(g.sum(1), 90) / g.sum(1).mean():.2f}, "
          f"max / mean {g.sum(1).max() / g.sum(1).mean():.2f}")
    # ordering: how are heavy tiles distributed along the launch order?
    q = np.array_split(nk, 8)
    print("   mean offsets per eighth of the launch order:", " ".join(f"{x.mean():.1f}" for x in q))
