"""Soak of the product loop: thousands of host-fed steps through ScanEngine (8 pipelines, the caller's stream one of them) with
alternating cloud sizes; every step's counts (rows, TP, FP, FN, TN) must equal the first run's exactly, the three f64 sums
(squared error, labels, labels squared: workgroup partials added with f64 atomics, so their last bit depends on the arrival order) to
1e-12, and device memory must not grow."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sps_amd import synthetic
from sps_amd.engine import ScanEngine
from sps_amd.models.models import SPSNet
net = bench.synthetic_weights(SPSNet(bench.CFG)).cuda().eval().freeze()
clouds = [torch.from_numpy(synthetic.make_scene(scan_seed=1 + i, n_azimuth=az)["batch"]).pin_memory() for i, az in enumerate((1750, 400, 1000, 1400))]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
eng = ScanEngine(net, 0, max_rows=max(len(c) for c in clouds), table_rows=N, stage_cols=6)
ref = eng.run_sequence(clouds)                       # one step per cloud: the reference sums
free0, _ = torch.cuda.mem_get_info()
t0 = time.perf_counter()
eng.reset_table(N)
for i in range(N):
    eng.submit(clouds[i % len(clouds)], 1)
sums = eng.finish().cpu().numpy()
dt = time.perf_counter() - t0
bad = sum(int(not (np.array_equal(sums[i, :5], ref[i % len(clouds), :5]) and np.allclose(sums[i, 5:], ref[i % len(clouds), 5:], rtol=1e-12, atol=0)))
          for i in range(N))
free1, _ = torch.cuda.mem_get_info()
print(f"{N} host-fed steps on {len(eng.streams)} pipelines in {dt:.2f} s ({N / dt:.0f}/s, rows {[len(c) for c in clouds]}); {bad} steps differ from the "
      f"reference sums; device memory {(free0 - free1) / 2**20:+.1f} MiB")
sys.exit(1 if bad else 0)
