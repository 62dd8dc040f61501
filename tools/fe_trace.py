"""DIAGNOSTIC: per-workgroup wall-clock stamps of the two ranking launches (k_rank_points, k_rank_blocks_rows): builds a
PRIVATE copy of the library with -DSPS_FE_TRACE on the GPU box (loaded through $SPS_LIB; the product library is untouched).
  gpurun -- python tools/fe_trace.py"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(tempfile.mkdtemp(prefix="sps_fe_"), "libsps_fe.so")
os.environ["SPS_LIB"] = lib
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSPS_FE_TRACE", *sys.argv[1:], "-o", lib,
                       os.path.join(ROOT, "sps_amd/csrc/sps_hip.hip")], stderr=subprocess.DEVNULL)
import numpy as np, torch
from sps_amd import synthetic, _native
from sps_amd.models.models import SPSNet
import bench
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
b = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
for _ in range(6):
    net(b)
torch.cuda.synchronize()
n = 4096
buf = (C.c_ulonglong * (8 * n))()
fn = _native.lib.sps_debug_fe_trace
fn.argtypes = [C.c_void_p, C.c_int]
_native.check(fn(buf, n))
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
names = ["start", "ticket", "published", "lookback", "ranked", "tocc", "ancestors/rows"]
for lo, hi, title in ((0, 1024, "k_rank_points"), (1024, 4096, "k_rank_blocks_rows")):
    tt = t[lo:hi]
    tt = tt[tt[:, 0] > 0]
    t0 = tt[:, 0].min()
    print(title, "workgroups", len(tt))
    for k, nm in enumerate(names):
        v = (tt[:, k] - t0) / 100.0
        v = v[tt[:, k] > 0]
        if len(v): print(f"  {nm:14s} n={len(v):4d} min {v.min():7.2f} median {np.median(v):7.2f} max {v.max():7.2f} us")

n = 16384
buf = (C.c_ulonglong * (2 * n))()
geom = (C.c_int * 8)()
fn = _native.lib.sps_debug_maps_trace
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
_native.check(fn(buf, n, geom))
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.int64)
nchunk, co = geom[0], list(geom)[1:7]
valid = t[:, 0] > 0
t0 = t[valid, 0].min()
print("k_maps: nchunk", nchunk, "chunk_off", co, "traced workgroups", int(valid.sum()))
def show(name, idx):
    idx = [i for i in idx if i < n and t[i, 0] > 0 and t[i, 1] > 0]
    if not idx: return
    st = (t[idx, 0] - t0) / 100.0; en = (t[idx, 1] - t0) / 100.0
    print(f"  {name:22s} n={len(idx):5d} start med {np.median(st):6.2f} max {st.max():6.2f} | end med {np.median(en):6.2f} max {en.max():6.2f} | dur med {np.median(en - st):6.2f} max {(en - st).max():6.2f} us")
for pos, sl in enumerate((0, 1, 2)):     # launch order of the time slices
    for l in range(5):
        show(f"nbr3 level {l} slice {sl}", range(pos * nchunk + co[l], pos * nchunk + co[l + 1]))
for f in range(4):
    show(f"stride maps {f}->{f+1}", range(3 * nchunk + co[f], 3 * nchunk + co[f + 1]))

buf = (C.c_ulonglong * (2 * n))()
fn = _native.lib.sps_debug_link_trace
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
_native.check(fn(buf, n, geom))
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.int64)
gb, co = geom[0], list(geom)[1:7]
valid = t[:, 0] > 0
t0 = t[valid, 0].min()
print("k_link_adj: gb", gb, "adjacency chunk offsets", co, "traced workgroups", int(valid.sum()))
for l in range(4):
    show(f"links level {l}->{l+1}", range(l * gb, (l + 1) * gb))
for i in range(5):
    show(f"adjacency level {4 - i}", range(4 * gb + co[i], 4 * gb + co[i + 1]))
