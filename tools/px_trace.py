"""Phase timeline of the pair-exact conv kernel (DIAGNOSTIC): private -DSPS_WAVE_TRACE build (loaded through $SPS_LIB, the
product library is never touched); k_conv_px stamps the shader clock at wave entry, after the counts, after the
fused downsample branch, after the chunk loop, after the workgroup barrier and after the epilogue.
  gpurun -- python tools/px_trace.py --layer block8.0.conv1 [--flags "-DSPS_PX_G=4"]"""
import argparse, ctypes as C, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="block8.0.conv1")
ap.add_argument("--flags", default="")
ap.add_argument("--mhz", type=float, default=100.0, help="s_memtime counts per microsecond")
args = ap.parse_args()
lib = os.path.join(tempfile.mkdtemp(prefix="sps_trace_"), "libsps_hip_trace.so")
os.environ["SPS_LIB"] = lib
try:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSPS_WAVE_TRACE",
                           *args.flags.split(), "-o", lib, os.path.join(ROOT, "sps_amd/csrc/sps_hip.hip")],
                          stderr=subprocess.DEVNULL)
    os.environ["SPS_TRACE_LAYER"] = args.layer
    import numpy as np, torch
    from sps_amd import synthetic, _native
    from sps_amd.models.models import SPSNet
    import bench
    net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
    b = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
    for _ in range(6):
        net(b)
    torch.cuda.synchronize()
    n = 16384
    buf = (C.c_ulonglong * (8 * n))()
    fn = _native.lib.sps_debug_px_trace
    fn.argtypes = [C.c_void_p, C.c_int]
    _native.check(fn(buf, n))
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    nk = t[:, 7] >> 32
    t0 = t[:, 0].min()
    us = lambda x: x / args.mhz
    pct = lambda x: " ".join(f"{np.percentile(x, p):7.2f}" for p in (0, 10, 50, 90, 99, 100))
    print(f"layer {args.layer}: {len(t)} waves, span {us(t[:, 6].max() - t0):.2f} us, chunks per supertile {pct(nk)}")
    print("entry (since first wave)            [us] p0 p10 p50 p90 p99 p100:", pct(us(t[:, 0] - t0)))
    names = ["count + chunk count + zero", "fused downsample branch", "-",
             "chunk loop", "wait at the barrier", "epilogue"]
    for i, nm in enumerate(names):
        print(f"{nm:36s}[us]", pct(us(t[:, i + 1] - t[:, i])))
    print(f"{'lifetime':36s}[us]", pct(us(t[:, 6] - t[:, 0])))
    edges = np.linspace(0, us(t[:, 6].max() - t0), 21)
    s, e = us(t[:, 0] - t0), us(t[:, 6] - t0)
    print("resident waves at", " ".join(f"{x:.0f}" for x in edges), "us:\n  ", [int(((s <= x) & (e > x)).sum()) for x in edges])
finally:
    shutil.rmtree(os.path.dirname(lib), ignore_errors=True)
