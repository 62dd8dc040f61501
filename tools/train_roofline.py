"""Roofline fractions of the training-step kernels from a rocprofv3 kernel-stats CSV of tools/train_timing.py.
usage: train_roofline.py <kernel_stats_train.csv> <bench.json of the same build (for the config-2 counts)>"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sps_amd import roofline as R

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous namespace" in r["Name"]]
steps = [int(r["Calls"]) for r in rows if "k_dlogit_accum" in r["Name"]][0]
us = lambda pred: sum(float(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / steps / 1e3
calls = lambda pred: sum(int(r["Calls"]) for r in rows if pred(r["Name"])) / steps
c = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["config"]
w = R.algorithmic_work(c["rows"], c["voxels_per_level"], c["pairs_3x3x3x3_per_level"], c["pairs_5x5x5x1"])
V = c["voxels_per_level"]
f = b = g = 0
for name, K, cin, cout, lin, lout, kind in R.LAYERS:
    if name in ("conv0p1s1", "final"):
        continue                                  # their weight gradients have kernels of their own
    P = w["per_layer"][name]["pairs"]
    f += 2 * P * cin * cout
    b += 4 * (V[lin] * cin + V[lout] * cout + K * cin * cout) + (8 * P if K > 1 else 0)
    g += 4 * P * (cin + cout)
is_wgrad = lambda n: "k_wgrad<" in n or "k_wgrad(" in n     # (k_wgrad<AW, BW> since round 4; the reduce kernels have other names)
t = us(is_wgrad) * 1e-6
tc = us(lambda n: "k_conv<" in n or "k_upconv" in n) * 1e-6
tb = us(lambda n: "k_bn_" in n)
print(f"kernel time per training step: {us(lambda n: True):.0f} us in {calls(lambda n: True):.0f} launches")
print(f"k_wgrad: {t * 1e6:.0f} us in {calls(is_wgrad):.0f} launches for {f / 1e9:.2f} GFLOP / {b / 1e6:.0f} MB algorithmic -> "
      f"{f / t / 1e12:.2f} TFLOP/s = {f / t / 1e12 / R.MFMA_F32_PEAK_TFLOPS:.3f} of the f32-MFMA peak, {b / t / 1e9:.0f} GB/s = "
      f"{b / t / 1e9 / R.HBM_PEAK_GBS:.3f} of the HBM roofline; gathered operand rows {g / 1e9:.2f} GB = {g / t / 1e12:.2f} TB/s through the L1s")
print(f"forward + data-gradient convolutions (k_conv, k_upconv): {tc * 1e6:.0f} us for 2 x {w['flops'] / 1e9:.2f} GFLOP -> "
      f"{2 * w['flops'] / tc / 1e12 / R.MFMA_F32_PEAK_TFLOPS:.3f} of the f32-MFMA peak")
print(f"BatchNorm kernels: {tb:.0f} us in {calls(lambda n: 'k_bn_' in n):.0f} launches ({tb / max(calls(lambda n: 'k_bn_' in n), 1):.1f} us each: launch floor)")
