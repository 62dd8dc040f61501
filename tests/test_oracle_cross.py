"""The two restatements of the oracle (numpy and C) against each other, and the weight-blob layout of
the C oracle against the native library's (they are written independently)."""
import numpy as np

from oracle import c_oracle, sps_oracle as O
from sps_amd import synthetic


def test_c_and_numpy_oracles_agree():
    p = O.random_params(seed=3)
    batch = synthetic.small_scene(seed=1, n_scan=2500)
    ref, info = O.sps_forward(p, batch[:, :5], 0.1)
    blob = c_oracle.pack_blob(p)
    for threads in (1, 4):
        s, ci = c_oracle.forward(blob, batch[:, :5], 0.1, nthreads=threads)
        np.testing.assert_array_equal(ci["voxels"], info["voxels"])
        np.testing.assert_array_equal(ci["inverse"], info["inverse"])
        assert ci["level_counts"] == [len(info["cm"].coords[1 << l]) for l in range(5)]
        np.testing.assert_allclose(ci["logits"], info["logits"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(s, ref, rtol=0, atol=5e-5)


def test_blob_layouts_agree():
    from sps_amd import _native        # loading the library needs no GPU
    assert list(c_oracle.layout()) == list(_native.weight_layout())
    names = {n for n, _, _ in c_oracle.layout()}
    assert set(O.random_params(0).keys()) == names


def test_c_oracle_empty_and_single():
    p = O.random_params(seed=3)
    blob = c_oracle.pack_blob(p)
    s, info = c_oracle.forward(blob, np.zeros((0, 5), np.float32), 0.1)
    assert s.shape == (0,) and info["level_counts"] == [0] * 5
    one = np.array([[0, 1.23, -4.56, 0.78, 1]], np.float32)
    s, _ = c_oracle.forward(blob, one, 0.1)
    ref, _ = O.sps_forward(p, one, 0.1)
    np.testing.assert_allclose(s, ref, atol=1e-6)
