"""Probe: partition the GPU with CU-masked streams (hipExtStreamCreateWithCUMask) -- P partitions of 256/P CUs,
k streams each -- and measure pipelined scans/s.  Kernels of this path are latency bound at low pipe utilisation,
so several forwards on disjoint CU sets might overlap better than on the shared wave slots."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet
hip = C.CDLL("libamdhip64.so")
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
x = torch.from_numpy(synthetic.make_scene(scan_seed=1)["batch"]).cuda()
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs:", NCU)

def masked_stream(bits):
    words = [0] * ((NCU + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * len(words))(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def run(streams, K=600):
    for s in streams:
        with torch.cuda.stream(s):
            net(x); net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % len(streams)]):
            net(x)
    torch.cuda.synchronize()
    return K / (time.perf_counter() - t0)

print("baseline 23 plain streams:", round(run([torch.cuda.Stream() for _ in range(23)])))
for P in (2, 4, 8):
    for mode in ("contiguous", "interleaved"):
        for k in (1, 2, 3):
            per = NCU // P
            streams = []
            for p in range(P):
                bits = list(range(p * per, (p + 1) * per)) if mode == "contiguous" else list(range(p, NCU, P))
                streams += [masked_stream(bits) for _ in range(k)]
            # interleave so consecutive scans go to different partitions
            order = [streams[p * k + j] for j in range(k) for p in range(P)]
            try:
                print(f"P={P} ({per} CUs each, {mode}), {k} stream(s) per partition: {run(order):.0f} scans/s", flush=True)
            except Exception as e:
                print("failed:", e)
