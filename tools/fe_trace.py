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
