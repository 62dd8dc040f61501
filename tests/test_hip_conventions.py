"""GPU tests of the ME-convention switchboard: for every alternative of sps_amd/conventions.py the HIP path with the
weights PACKED under option X equals the oracle computing GEOMETRICALLY under option X -- and differs from the default
reading of the same stored parameters (the option is not vacuous).  No kernel knows about conventions: the blob
permutation at packing time is the whole mechanism (inference: NativeBackboneModule.device_weights; training:
_TrainForward + the inverse permutation of the gradient)."""
import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from oracle import train_oracle as T
from sps_amd import conventions as CV, synthetic
from tests.helpers import CFG, net_from_params, state_dict_from_params, straddle_params
from tests.test_hip_parity import ctx, get_feature, get_voxels, match_rows

pytestmark = pytest.mark.gpu
VS = CFG["MODEL"]["VOXEL_SIZE"]
SINGLE = [CV.MEConventions(**{k: vals[1]}) for k, vals in CV.OPTIONS.items()]
ALL_FLIPPED = CV.MEConventions(**{k: vals[1] for k, vals in CV.OPTIONS.items()})
TAP_LEVEL = {"out_p1": 0, "block1": 1, "block2": 2, "block3": 3, "block4": 4, "block5": 3, "block6": 2, "block7": 1, "block8": 0}


@pytest.fixture(scope="module")
def scene():
    return synthetic.small_scene(seed=21, n_scan=1800)


@pytest.fixture(scope="module")
def params(scene):
    return straddle_params(O.random_params(seed=5), scene)


@pytest.mark.parametrize("cv", SINGLE + [ALL_FLIPPED], ids=lambda c: c.describe())
def test_hip_packed_under_option_equals_oracle_under_option(cv, params, scene):
    net = net_from_params(params)
    net.model.set_me_conventions(cv)
    net = net.cuda().eval().freeze()
    dev = torch.from_numpy(scene).cuda()
    s = net(dev).cpu().numpy()
    want, info = O.sps_forward(params, scene[:, :5], VS, keep=True, cv=cv)
    counts = ctx().level_counts()
    perm = [match_rows(get_voxels(l, counts[l]), info["cm"].coords[1 << l]) for l in range(5)]
    for name, feat in info["inter"].items():                      # every stage of the network, not just the end
        np.testing.assert_allclose(get_feature(name), feat[perm[TAP_LEVEL[name]]], rtol=2e-4, atol=2e-4, err_msg=name)
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-4)
    # the same stored parameters read with the default conventions: a different network
    plain = net_from_params(params).cuda().eval().freeze()
    s0 = plain(dev).cpu().numpy()
    want0, _ = O.sps_forward(params, scene[:, :5], VS)
    np.testing.assert_allclose(s0, want0, rtol=0, atol=1e-4)
    assert np.max(np.abs(s0 - want)) > 1e-3 and np.mean(np.abs(s0 - want) > 1e-4) > 0.5, "the option is vacuous on this scene"
    # switching a live module back re-packs the blob
    net.model.set_me_conventions(None)
    np.testing.assert_allclose(net(dev).cpu().numpy(), want0, rtol=0, atol=1e-4)


def test_lin_kernels_stored_out_in_load_and_run(params, scene):
    """A checkpoint whose 1x1 kernels are stored [C_out, C_in]: same network as the canonical one with transposed tensors."""
    cv = CV.MEConventions(lin_layout="out_in")
    stored = {k: (np.ascontiguousarray(np.asarray(v).T) if (k.endswith(".kernel") and np.asarray(v).ndim == 2) else v)
              for k, v in params.items()}
    from sps_amd.models.models import SPSNet
    net = SPSNet({**CFG, "MODEL": {"VOXEL_SIZE": VS, "ME_CONVENTIONS": cv.describe()}})
    net.load_state_dict(state_dict_from_params(stored))
    net = net.cuda().eval().freeze()
    dev = torch.from_numpy(scene).cuda()
    want, _ = O.sps_forward(params, scene[:, :5], VS)              # canonical parameters, canonical reading
    np.testing.assert_allclose(net(dev).cpu().numpy(), want, rtol=0, atol=1e-4)


@pytest.mark.timeout(900)
def test_training_gradients_under_an_option_match_the_autograd_oracle():
    """The training step sees the permuted copy of the flat parameter tensor and hands the gradient back through the
    inverse permutation: gradients in the STORED layout == autograd of the oracle computing under the option."""
    cv = CV.MEConventions(offset_order="t_fastest", even_kernel_order="descending", transpose_index="mirrored",
                          odd_kernel_sign="minus", lin_layout="out_in")
    batch = synthetic.small_scene(seed=3, n_scan=700)
    params = O.random_params(seed=4)
    loss_ref, scores_ref, grads_ref, _ = T.train_step(params, batch, VS, cv=cv)
    net = net_from_params(params)
    net.model.set_me_conventions(cv)
    net = net.cuda().train()
    out = net.training_step(torch.from_numpy(batch).cuda(), 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    assert float(out["loss"].detach()) == pytest.approx(loss_ref, rel=2e-5)
    grads = {k.replace("model.MinkUNet.", ""): p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}
    bad = {}
    for name, want in grads_ref.items():
        got = grads[name].reshape(want.shape)
        e = float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-12))
        if not e <= 2e-3:
            bad[name] = e
    assert not bad, f"gradient mismatch (relative to the tensor's max): {bad}"
    # and not by accident: against the default-convention oracle the permuted kernels' gradients are far off
    _, _, grads_plain, _ = T.train_step(params, batch, VS)
    k = "block1.0.conv1.kernel"
    assert np.max(np.abs(grads[k] - grads_plain[k])) > 0.05 * np.max(np.abs(grads_plain[k]))


@pytest.mark.timeout(900)
def test_convention_probe_finds_the_convention_a_checkpoint_was_written_in(scene):
    """tools/convention_probe.py: labels = the scores of a teacher that computes under a hidden option set; of the 32
    readings of its checkpoint only the hidden one reproduces them (R2 = 1), and it is ranked first."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import convention_probe
    hidden = CV.MEConventions(offset_order="t_fastest", transpose_index="mirrored")
    teacher = O.random_params(seed=9)
    scores, _ = O.sps_forward(teacher, scene[:, :5], VS, cv=hidden)
    batch = scene.copy()
    batch[:, 5] = scores
    res = convention_probe.probe(state_dict_from_params(teacher), CFG, torch.from_numpy(batch).cuda())
    assert len(res) == 32
    best_cv, best = res[0]
    assert best_cv == hidden and best["r2"] > 0.9999, (best_cv, best)
    assert res[1][1]["r2"] < 0.99, [(c.describe(), m["r2"]) for c, m in res[:4]]
