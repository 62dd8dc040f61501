"""Issue / L1 budget of the 24 k_conv launches for the config-2 scene (GPU): per layer the executed
(tile, offset) pairs, MFMA instructions, MFMA time if spread perfectly over 1024 SIMDs, and the L1 bytes
of the A gathers and the weight (B) loads at 64 B/clk/CU.  Compares with the measured stage times."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sps_amd import synthetic, _native
from sps_amd.models.models import SPSNet, get_context
import bench

CLK = 2.4e9
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
b = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
ctx = get_context(0)
for _ in range(5):
    net(b)
ctx.profile_enable(True)
net(b); torch.cuda.synchronize()
stage = dict(ctx.profile_read())
ctx.profile_enable(False)
V = ctx.level_counts()
present = {}
for which in range(5):
    n = C.c_int64()
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, which, None, C.byref(n)))
    m = torch.empty((n.value, 4), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_tile_masks(ctx.handle, which, m.data_ptr(), C.byref(n)))
    present[which] = np.unpackbits(m.cpu().numpy().view(np.uint8), axis=1).sum(1).astype(np.int64)
# (name, level, cin, cout, ds_cin, ntw)   3^4 layers only + stride layers (K = 8: every tile has <= 8 offsets)
P = (8, 16, 32, 64, 64, 32, 16, 8)
layers = [("block1.0.conv1", 1, 8, 8, 0), ("block1.0.conv2", 1, 8, 8, 0), ("block2.0.conv1", 2, 8, 16, 0),
          ("block2.0.conv2", 2, 16, 16, 8), ("block3.0.conv1", 3, 16, 32, 0), ("block3.0.conv2", 3, 32, 32, 16),
          ("block4.0.conv1", 4, 32, 64, 0), ("block4.0.conv2", 4, 64, 64, 32), ("block5.0.conv1", 3, 96, 64, 0),
          ("block5.0.conv2", 3, 64, 64, 96), ("block6.0.conv1", 2, 48, 32, 0), ("block6.0.conv2", 2, 32, 32, 48),
          ("block7.0.conv1", 1, 24, 16, 0), ("block7.0.conv2", 1, 16, 16, 24), ("block8.0.conv1", 0, 16, 8, 0),
          ("block8.0.conv2", 0, 8, 8, 16)]
tot_mfma = tot_l1 = tot_ms = 0.0
print(f"{'layer':16s} {'tiles':>6s} {'pairs':>8s} {'MFMA':>9s} {'mfma_us':>8s} {'L1_MB':>7s} {'l1_us':>6s} {'meas_us':>8s}")
for name, lvl, cin, cout, ds, in layers:
    pc = present[lvl]
    upk = cin // 4
    nt = (cout + 15) // 16
    ntw = nt if lvl <= 1 else 1
    units = pc * upk + ds // 4
    iters = ((units + 3) // 4).sum()
    mfma = iters * 4 * nt
    mfma_us = mfma * 32 / (1024 * CLK) * 1e6
    l1 = iters * 1024 * (nt // ntw) + iters * 1024 * nt
    l1_us = l1 / (64 * 256 * CLK) * 1e6
    meas = stage.get(name, 0) * 1000
    tot_mfma += mfma_us; tot_l1 += l1_us; tot_ms += meas
    print(f"{name:16s} {len(pc):6d} {int(pc.sum()):8d} {int(mfma):9d} {mfma_us:8.1f} {l1 / 1e6:7.1f} {l1_us:6.1f} {meas:8.1f}")
print(f"total 3^4 layers: MFMA {tot_mfma:.0f} us, L1 {tot_l1:.0f} us, measured serial {tot_ms:.0f} us; levels {V}")
