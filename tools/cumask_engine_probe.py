"""Do the engine's pipelines run better on CU partitions of their own?  (DIAGNOSTIC, GPU)
hipExtStreamCreateWithCUMask applies a mask symmetrically to the 8 XCCs (tools/microbench/cumask_probe.hip: mask bit j = CU j // 8 of
XCC j % 8; 32 CUs per XCC): P partitions = every pipeline gets 32 / P CUs of EVERY XCC; pipeline i uses partition i % P.  The side
streams of ScanEngine are replaced by CU-masked ones (the caller's stream, one of the pipelines, stays unmasked unless --mask-main).
Prints the resident-input rate of 300 steps for P = 1 (no masks), 2, 4."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sps_amd import synthetic
from sps_amd.engine import ScanEngine
from sps_amd.models.models import SPSNet

hip = C.CDLL("libamdhip64.so")


def masked_stream(part, parts):
    mask = (C.c_uint32 * 8)()
    per = 32 // parts
    for cu in range(part * per, (part + 1) * per):
        for xcc in range(8):
            j = cu * 8 + xcc
            mask[j >> 5] |= 1 << (j & 31)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def run(parts, steps=300):
    counter = [0]
    real = torch.cuda.Stream

    def fake(*a, **k):
        counter[0] += 1
        return masked_stream(counter[0] % parts, parts) if parts > 1 else real(*a, **k)

    import sps_amd.engine as E
    E.torch.cuda.Stream = fake
    try:
        eng = ScanEngine(net, 0, streams=8, max_rows=max(len(b) for b in batches), table_rows=steps)
    finally:
        E.torch.cuda.Stream = real
    dev = [torch.from_numpy(b).cuda() for b in batches]
    for rep in range(2):
        eng.reset_table(steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.submit(dev[i % len(dev)], 1, row=i)
        eng.finish()
        dt = time.perf_counter() - t0
    return steps / dt


net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
mp = synthetic.build_map(n_azimuth=1750)
batches = [synthetic.make_scene(scan_seed=1 + 100 * i, n_azimuth=1750, voxel_size=0.1, map_points=mp)["batch"] for i in range(4)]
for parts in (1, 2, 4, 1, 2, 4):
    print(f"{parts} CU partition(s) per XCC: {run(parts):.0f} scans/s (resident inputs, 8 pipelines)")
