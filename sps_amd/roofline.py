"""Algorithmic work of one forward: the single definition of B_alg / F_alg used by
bench.py's ``roofline`` object and by DESIGN.md.  Computed from the run's exact integer counts.

  B_vox  = 20 N (read coords) + 4 N (inverse map) + 20 V1 (voxel coords) + 4 N (scores)
  B_maps = sum over the 10 distinct kernel maps [20 V_out + 8 P] + sum over 4 stride maps 28 V_fine
  B_conv = sum over the 33 convs 4 (V_in C_in + V_out C_out + K C_in C_out) + (8 P if K > 1)
  B_res  = sum over the 8 BasicBlocks 4 V C_out
  F_alg  = sum over the 33 convs 2 P C_in C_out      (P = V for 1x1 and stride/transposed layers)
BN / ReLU / cat / sigmoid count 0 (fused); hash probe traffic is excluded.
"""
from __future__ import annotations

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3

# (name, K, C_in, C_out, level_in, level_out, map kind)
LAYERS = [("conv0p1s1", 125, 1, 8, 0, 0, "k5")]
_enc = [8, 8, 16, 32, 64]
for _i, _n in enumerate(("conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2")):
    _cin, _cout, _l = _enc[_i], _enc[_i + 1], _i + 1
    LAYERS.append((_n, 8, _cin, _cin, _l - 1, _l, "down"))
    LAYERS.append((f"block{_l}.0.conv1", 81, _cin, _cout, _l, _l, "k3"))
    LAYERS.append((f"block{_l}.0.conv2", 81, _cout, _cout, _l, _l, "k3"))
    if _cin != _cout:
        LAYERS.append((f"block{_l}.0.downsample.0", 1, _cin, _cout, _l, _l, "lin"))
_dec = [(64, 64, 96), (64, 32, 48), (32, 16, 24), (16, 8, 16)]      # (C_in of convtr, C_out, concat width)
for _i, _n in enumerate(("convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2")):
    _cin, _cout, _cat = _dec[_i]
    _l = 3 - _i
    LAYERS.append((_n, 8, _cin, _cout, _l + 1, _l, "up"))
    LAYERS.append((f"block{5 + _i}.0.conv1", 81, _cat, _cout, _l, _l, "k3"))
    LAYERS.append((f"block{5 + _i}.0.conv2", 81, _cout, _cout, _l, _l, "k3"))
    LAYERS.append((f"block{5 + _i}.0.downsample.0", 1, _cat, _cout, _l, _l, "lin"))
LAYERS.append(("final", 1, 8, 1, 0, 0, "lin"))
assert len(LAYERS) == 33


def layer_pairs(kind, lin, lout, V, pairs3, pairs5):
    if kind == "k5":
        return pairs5
    if kind == "k3":
        return pairs3[lout]
    if kind == "down":
        return V[lin]          # every fine voxel feeds exactly one parent through one offset
    if kind == "up":
        return V[lout]         # every fine voxel receives exactly one term
    return V[lout]


def algorithmic_work(n_points: int, V, pairs3, pairs5) -> dict:
    """V: voxels per level [5]; pairs3: total 3^4 pairs per level [5]; pairs5: total 5x5x5x1 pairs."""
    per_layer = {}
    b_conv = f_alg = 0
    for name, K, cin, cout, lin, lout, kind in LAYERS:
        P = layer_pairs(kind, lin, lout, V, pairs3, pairs5)
        b = 4 * (V[lin] * cin + V[lout] * cout + K * cin * cout) + (8 * P if K > 1 else 0)
        f = 2 * P * cin * cout
        per_layer[name] = dict(bytes=b, flops=f, pairs=P)
        b_conv += b
        f_alg += f
    block_out = [(1, 8), (2, 16), (3, 32), (4, 64), (3, 64), (2, 32), (1, 16), (0, 8)]
    b_res = sum(4 * V[l] * c for l, c in block_out)
    b_maps = 20 * V[0] + 8 * pairs5
    b_maps += sum(20 * V[l] + 8 * pairs3[l] for l in range(5))
    b_maps += sum(20 * V[l + 1] + 8 * V[l] for l in range(4))
    b_maps += sum(28 * V[l] for l in range(4))
    b_vox = 20 * n_points + 4 * n_points + 20 * V[0] + 4 * n_points
    return dict(bytes=b_vox + b_maps + b_conv + b_res, flops=f_alg, b_vox=b_vox, b_maps=b_maps, b_conv=b_conv,
                b_res=b_res, per_layer=per_layer)
