"""GPU fuzz: random small clouds (sizes, extents, duplicates, batches, time slices) alternating with a larger one on
the SAME context -- exercises stale tile masks / neighbour entries of a previous, bigger forward, partial last tiles,
empty levels and the split-K workgroup mapping -- against the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from sps_amd import synthetic
from tests.helpers import net_from_params

pytestmark = pytest.mark.gpu


def _random_cloud(rng):
    n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 2500]))
    extent = float(rng.choice([0.3, 1.0, 3.0, 12.0]))
    xyz = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    if rng.random() < 0.5:
        xyz[:, 2] = np.round(xyz[:, 2] * 2) / 2 * 0.1          # layered
    if rng.random() < 0.3 and n > 4:
        xyz[n // 2:] = xyz[: n - n // 2]                        # exact duplicates
    nb = int(rng.choice([1, 1, 2, 4]))
    b = rng.integers(0, nb, n).astype(np.float32)
    t = rng.choice(np.array([0.0, 1.0], np.float32), n) if rng.random() < 0.8 else rng.integers(-2, 3, n).astype(np.float32)
    return np.concatenate([b[:, None], xyz, t[:, None]], 1).astype(np.float32)


def test_fuzz_small_clouds_after_big_ones():
    params = O.random_params(seed=11)
    net = net_from_params(params).cuda().eval().freeze()
    big = torch.from_numpy(synthetic.make_scene(scan_seed=3, n_azimuth=600)["batch"]).cuda()
    rng = np.random.default_rng(2024)
    worst = 0.0
    for i in range(24):
        if i % 6 == 0:
            net(big)                                            # leaves large structures behind
        c = _random_cloud(rng)
        got = net.model(torch.from_numpy(c).cuda()).cpu().numpy()
        want, _ = O.sps_forward(params, c, 0.1)
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        assert err < 1e-4, f"case {i}: n={len(c)} err={err}"
    torch.cuda.synchronize()
    assert worst < 1e-4
