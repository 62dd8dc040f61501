"""Thread-scaling probe of the C oracle on the current host (informs bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle, sps_oracle as O
from sps_amd import synthetic
sc = synthetic.make_scene(scan_seed=1)
b = np.ascontiguousarray(sc["batch"][:, :5])
blob = c_oracle.pack_blob(O.random_params(0))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 4, 8, 16, 32, 64):
    if th > (os.cpu_count() or 1): break
    c_oracle.forward(blob, b, 0.1, nthreads=th, want_details=False)
    t = time.time(); s, info = c_oracle.forward(blob, b, 0.1, nthreads=th, want_details=False); dt = time.time() - t
    print(th, "threads", round(dt, 3), "s", [round(x, 3) for x in info["timings"]], flush=True)
