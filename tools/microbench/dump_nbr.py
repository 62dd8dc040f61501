"""Dump the level-`L` 3^4 neighbour table + tile masks of the config-2 scene for gather_replay."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sps_amd import synthetic, _native
from sps_amd.models.models import SPSNet, get_context
import bench
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
b = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
net(b); torch.cuda.synchronize()
ctx = get_context(0)
V = ctx.level_counts()[level]
n = C.c_int64()
_native.check(_native.lib.sps_get_tile_masks(ctx.handle, level, None, C.byref(n)))
m = torch.empty((n.value, 4), dtype=torch.int32, device="cuda")
_native.check(_native.lib.sps_get_tile_masks(ctx.handle, level, m.data_ptr(), C.byref(n)))
nbr = torch.empty((81, V), dtype=torch.int32, device="cuda")
_native.check(_native.lib.sps_get_nbr(ctx.handle, level, nbr.data_ptr()))
ldn = (V + 15) // 16 * 16
full = np.full((81, ldn), -1, np.int32)
full[:, :V] = nbr.cpu().numpy()
# entries whose (tile, k) mask bit is clear are never written by the map builder: make them -1
mm = m.cpu().numpy().view(np.uint32)
allbits = np.unpackbits(mm.view(np.uint8), axis=1, bitorder="little").astype(bool)        # [tiles, 128]
bits = allbits[:, [32 * (k // 27) + k % 27 for k in range(81)]]                               # one word per time slice
full = np.where(np.repeat(bits.T, 16, axis=1)[:, :ldn], full, -1)
with open("/tmp/nbr.bin", "wb") as f:
    np.array([V, ldn, n.value], np.int64).tofile(f)
    full.astype(np.int32).tofile(f)
    packed = np.zeros((len(mm), 128), bool); packed[:, :81] = bits
    np.packbits(packed, axis=1, bitorder="little").view(np.uint32).tofile(f)   # contiguous 81-bit layout for gather_replay
print("level", level, "V", V, "tiles", n.value, "pairs", int((full >= 0).sum()))
