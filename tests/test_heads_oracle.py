"""CPU checks of the oracle's baseline-head restatement: 4DMOS / MapMOS run the same
backbone, so what is new is the voxel-mean feature, the k-channel `final` and the shift invariance along t
that the product path relies on when it re-bases a long-running scan index."""
import numpy as np

from oracle import sps_oracle as o
from sps_amd.synthetic import small_scene


def _cloud(seed, n, t_values):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    xyz[:, 2] *= 0.1                                   # a slab: neighbours exist
    t = rng.choice(np.asarray(t_values, dtype=np.float32), n)
    return np.concatenate([np.zeros((n, 1), np.float32), xyz, t[:, None]], 1).astype(np.float32)


def test_voxel_mean_kat():
    f = np.array([1.0, 2.0, 4.0, 8.0, 1.5], np.float32)
    inv = np.array([0, 1, 0, 1, 2])
    m = o.voxel_mean(f, inv, 3)
    assert m.shape == (3, 1)
    np.testing.assert_array_equal(m[:, 0], np.array([2.5, 5.0, 1.5], np.float32))


def test_mapmos_features_match_reference_formula():
    # mapmos.py:65-71 evaluated by hand: scan index 1, map index 0 -> features 1 and 2 (mapmos_node.py:92-93)
    idx = np.array([1, 1, 0, 0, 0], np.float32)
    np.testing.assert_array_equal(o.mapmos_features(idx), np.array([1, 1, 2, 2, 2], np.float32))
    np.testing.assert_array_equal(o.mapmos_features(np.ones(4, np.float32)), np.ones(4, np.float32))


def test_constant_half_feature_equals_sps_logits():
    # with features == 0.5 everywhere and a 1-channel head, head_forward is SPSModel.forward before the sigmoid
    p = o.random_params(3)
    batch = small_scene(seed=5, n_scan=600)
    scores, info = o.sps_forward(p, batch[:, :5], 0.1)
    logits, _ = o.head_forward(p, batch[:, :5], 0.1, features=np.full(len(batch), 0.5, np.float32))
    np.testing.assert_array_equal(o.sigmoid(logits[:, 0]), scores)


def test_time_shift_invariance():
    # every stride is [2,2,2,1] and no layer sees absolute t: shifting all t by an integer changes nothing
    p = o.random_params(1, out_channels=3)
    c = _cloud(0, 800, range(10))
    a, _ = o.head_forward(p, c, 0.2)
    c2 = c.copy()
    c2[:, 4] += 1000.0
    b, _ = o.head_forward(p, c2, 0.2)
    assert a.shape == (800, 3)
    np.testing.assert_array_equal(a, b)


def test_head_columns_are_independent_linear_maps():
    # column j of a k-channel head == a 1-channel head carrying final.kernel[:, j] / final.bias[:, j]
    p3 = o.random_params(2, out_channels=3)
    c = _cloud(1, 500, (0, 1, 2))
    full, _ = o.head_forward(p3, c, 0.2)
    for j in range(3):
        p1 = dict(p3)
        p1["final.kernel"] = p3["final.kernel"][:, j:j + 1].copy()
        p1["final.bias"] = p3["final.bias"][:, j:j + 1].copy()
        one, _ = o.head_forward(p1, c, 0.2)
        np.testing.assert_allclose(one[:, 0], full[:, j], rtol=0, atol=1e-6)


def test_head_blob_layout_matches_parameter_containers():
    """The native k-channel layouts (sps_head_tensor_info) name exactly the tensors of the Python containers, with
    the reference's state_dict keys and sizes (final.kernel [8,k], final.bias [1,k]); no GPU needed."""
    from sps_amd import _native
    from sps_amd.models.baselines import MapMOSNet, MOS4DNet
    for model, oc in ((MOS4DNet(0.2), 3), (MapMOSNet(0.1), 1)):
        sd = {k: v for k, v in model.MinkUNet.state_dict().items() if not k.endswith("num_batches_tracked")}
        layout = _native.weight_layout(oc)
        assert {n for n, _, _ in layout} == set(sd)
        off = 0
        for name, o, numel in layout:
            assert o == off and sd[name].numel() == numel, name
            off += numel
        assert off == _native.lib.sps_head_numel(oc)
        assert tuple(sd["final.kernel"].shape) == (8, oc) and tuple(sd["final.bias"].shape) == (1, oc)
    assert _native.lib.sps_head_numel(1) == _native.lib.sps_weights_numel()
    assert _native.lib.sps_head_numel(9) < 0            # out of range is an error, not a crash
