"""Per-wave timeline of one conv launch (DIAGNOSTIC): builds a PRIVATE copy of the library with -DSPS_WAVE_TRACE on the
GPU box (the bindings load $SPS_LIB; the product libsps_hip.so is never touched), records wall-clock stamps (100 MHz) at
wave entry / first tile / exit for the layer named by --layer, prints the distribution of start times and lifetimes.
  gpurun -- python tools/wave_trace.py --layer block8.0.conv1"""
import argparse, ctypes as C, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="block8.0.conv1")
ap.add_argument("--flags", default="")
args = ap.parse_args()
import tempfile
lib = os.path.join(tempfile.mkdtemp(prefix="sps_trace_"), "libsps_hip_trace.so")
os.environ["SPS_LIB"] = lib
try:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSPS_DIAG", "-DSPS_WAVE_TRACE",
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
    n = 32768
    buf = (C.c_ulonglong * (4 * n))()
    fn = _native.lib.sps_debug_wave_trace
    fn.argtypes = [C.c_void_p, C.c_int]
    _native.check(fn(buf, n))
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    t = t[t[:, 0] > 0]
    tiles = t[:, 3] >> 32
    units = t[:, 3] & 0xFFFFFFFF
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0          # us
    first = (t[:, 1] - t[:, 0]) / 100.0
    life = (t[:, 2] - t[:, 0]) / 100.0
    end = (t[:, 2] - t0) / 100.0
    w = tiles > 0
    pct = lambda x: " ".join(f"{np.percentile(x, p):6.2f}" for p in (0, 10, 50, 90, 99, 100))
    print(f"layer {args.layer}: {len(t)} waves recorded, {int(w.sum())} with tiles, span {end.max():.2f} us")
    print("start time  [us] p0 p10 p50 p90 p99 p100 :", pct(start))
    print("working waves: start                      :", pct(start[w]))
    print("working waves: entry -> first tile  [us]  :", pct(first[w]))
    print("working waves: lifetime             [us]  :", pct(life[w]))
    print("idle waves   : lifetime             [us]  :", pct(life[~w]) if (~w).any() else "-")
    print("units per working wave                    :", pct(units[w].astype(float)))
    # concurrency over time
    edges = np.linspace(0, end.max(), 23)
    conc = [int(((start[w] <= e) & (end[w] > e)).sum()) for e in edges]
    print("resident working waves at", " ".join(f"{e:.0f}" for e in edges), "us:\n  ", conc)
finally:
    shutil.rmtree(os.path.dirname(lib), ignore_errors=True)
