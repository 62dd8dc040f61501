"""CPU checks of the training oracle (oracle/train_oracle.py): its forward is the inference oracle's network with
train-mode BatchNorm, and its autograd gradients agree with central finite differences in float64."""
import numpy as np

from oracle import sps_oracle as O
from oracle import train_oracle as T
from sps_amd import synthetic

VS = 0.1


def test_train_oracle_matches_finite_differences():
    batch = synthetic.small_scene(seed=2, n_scan=150)
    p = O.random_params(seed=3)
    loss, scores, grads, stats = T.train_step(p, batch, VS)
    assert np.isfinite(loss) and scores.shape == (len(batch),) and len(grads) == 98
    rng = np.random.default_rng(0)
    for name in ("block8.0.conv2.kernel", "convtr5p8s2.kernel", "block3.0.downsample.0.kernel", "block2.0.norm1.bn.weight",
                 "conv0p1s1.kernel", "final.bias"):
        flat = np.asarray(p[name]).reshape(-1)
        j = int(rng.integers(0, flat.size))
        h = 1e-6
        vals = []
        for sgn in (+1, -1):
            q = dict(p)
            w = np.asarray(p[name], dtype=np.float64).copy()
            w.reshape(-1)[j] += sgn * h
            q[name] = w
            vals.append(T.train_step(q, batch, VS)[0])
        fd = (vals[0] - vals[1]) / (2 * h)
        an = grads[name].reshape(-1)[j]
        assert abs(fd - an) <= 1e-8 + 2e-3 * abs(fd), (name, j, fd, an)


def test_train_mode_batchnorm_statistics_and_eval_consistency():
    """With the batch statistics written into running_mean / running_var, the EVAL oracle (sps_oracle, float32)
    reproduces the train oracle's scores: same wiring, same maps, only BatchNorm's statistics differ."""
    batch = synthetic.small_scene(seed=4, n_scan=300)
    p = O.random_params(seed=5)
    _, scores, _, stats = T.train_step(p, batch, VS)
    q = dict(p)
    for bn, (mean, var, n) in stats.items():
        q[bn + ".bn.running_mean"] = mean.astype(np.float32)
        q[bn + ".bn.running_var"] = var.astype(np.float32)          # biased variance normalises in train mode
        assert n > 1 and (var >= 0).all()
    ref, _ = O.sps_forward(q, batch[:, :5], VS)
    np.testing.assert_allclose(ref, scores, rtol=0, atol=5e-5)
