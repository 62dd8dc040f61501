"""CPU tests of the ME-convention switchboard (sps_amd/conventions.py): the weight-blob PERMUTATION the product applies
is held against the oracle, which realises the same options in the GEOMETRY it enumerates (oracle/sps_oracle.py
kernel_offsets / lin_kernel / the mirrored transposed index) -- two independent implementations of every option."""
import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from sps_amd import _native, conventions as CV, synthetic
from tests.helpers import CFG, state_dict_from_params

VS = CFG["MODEL"]["VOXEL_SIZE"]
SINGLE = [CV.MEConventions(**{k: vals[1]}) for k, vals in CV.OPTIONS.items()]
ALL_FLIPPED = CV.MEConventions(**{k: vals[1] for k, vals in CV.OPTIONS.items()})


def test_option_table_matches_the_oracle_and_parses():
    assert {k: v[0] for k, v in CV.OPTIONS.items()} == O.CV_DEFAULTS          # same names, same defaults
    assert CV.DEFAULT.is_default and CV.parse(None) is CV.DEFAULT and CV.parse("") is CV.DEFAULT
    cv = CV.parse("offset_order=t_fastest, transpose_index=mirrored")
    assert cv == CV.MEConventions(offset_order="t_fastest", transpose_index="mirrored") and not cv.is_default
    assert CV.parse({"lin_layout": "out_in"}).lin_layout == "out_in"
    assert CV.parse(cv.describe()) == cv
    with pytest.raises(ValueError):
        CV.parse("offset_order=sideways")
    with pytest.raises(ValueError):
        CV.parse("kernel_shape=round")
    combos = CV.all_combinations()
    assert len(combos) == 32 and len(set(combos)) == 32 and combos[0].is_default


@pytest.mark.parametrize("cv", CV.all_combinations(), ids=lambda c: c.describe())
def test_kernel_index_map_agrees_with_the_oracles_geometry(cv):
    """W_internal[k] = W_ckpt[kperm[k]]  <=>  the offset the oracle gives checkpoint index kperm[k] under ``cv`` is the
    offset it gives index k under the defaults."""
    for kind, ksize in CV.KSIZE.items():
        for ts in (1, 4):
            theirs = O.kernel_offsets(ksize, ts, cv)
            if kind == "up" and cv.transpose_index == "mirrored":
                theirs = theirs[::-1]                                           # oracle: W[::-1] on the same map
            canon = O.kernel_offsets(ksize, ts)
            kperm = CV.kernel_index_map(kind, cv)
            assert sorted(kperm.tolist()) == list(range(len(canon)))
            np.testing.assert_array_equal(theirs[kperm], canon)
            scale = np.array([ts, ts, ts, 1])
            np.testing.assert_array_equal(CV.index_offsets(kind, cv) * scale,
                                          O.kernel_offsets(ksize, ts, cv)[::-1] if (kind == "up" and cv.transpose_index == "mirrored")
                                          else O.kernel_offsets(ksize, ts, cv))


def _unpack(blob, layout, like):
    return {name: blob[off: off + num].reshape(np.asarray(like[name]).shape) for name, off, num in layout}


@pytest.mark.parametrize("cv", SINGLE + [ALL_FLIPPED], ids=lambda c: c.describe())
def test_permuted_blob_under_default_equals_stored_blob_under_option(cv):
    """oracle(parameters as stored, options cv) == oracle(blob[perm] unpacked, defaults) up to the f32 summation order
    (the oracle adds the offsets in checkpoint index order), and the option is not vacuous: the stored parameters read
    with the defaults give different scores."""
    params = O.random_params(seed=3)
    layout = _native.weight_layout(1)
    blob = np.concatenate([np.asarray(params[name], np.float32).reshape(-1) for name, _, _ in layout])
    shapes = {n: (np.asarray(params[n]).shape if np.asarray(params[n]).ndim == 3 else (1,) + np.asarray(params[n]).shape)
              for n, _, _ in layout if n.endswith(".kernel")}
    perm = CV.blob_permutation(layout, shapes, cv)
    assert perm is not None and sorted(perm.tolist()) == list(range(len(blob)))
    inv = CV.inverse_permutation(perm)
    np.testing.assert_array_equal(blob[perm][inv], blob)
    batch = synthetic.small_scene(seed=4, n_scan=400)
    want, _ = O.sps_forward(params, batch[:, :5], VS, cv=cv)
    got, _ = O.sps_forward(_unpack(blob[perm], layout, params), batch[:, :5], VS)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    plain, _ = O.sps_forward(params, batch[:, :5], VS)
    assert np.max(np.abs(plain - want)) > 1e-4, "the option changed nothing"
    assert CV.blob_permutation(layout, shapes, CV.MEConventions()) is None


def test_module_carries_the_conventions_and_accepts_lin_kernel_shapes():
    from sps_amd.models.models import SPSNet
    cfg = {**CFG, "MODEL": {"VOXEL_SIZE": 0.1, "ME_CONVENTIONS": "lin_layout=out_in,offset_order=t_fastest"}}
    net = SPSNet(cfg)
    assert net.model.me_conventions == CV.MEConventions(offset_order="t_fastest", lin_layout="out_in")
    perm = net.model.blob_permutation()
    assert perm is not None and perm.shape == (_native.lib.sps_weights_numel(),)
    assert SPSNet(CFG).model.blob_permutation() is None                         # reference config: canonical
    # a checkpoint whose 1x1 kernels are [C_out, C_in] (or 3-D [1, C_in, C_out]) loads strictly; the memory is kept
    params = O.random_params(seed=1)
    sd = state_dict_from_params(params)
    k = "model.MinkUNet.block2.0.downsample.0.kernel"                            # [8, 16]
    stored = torch.arange(128, dtype=torch.float32).reshape(16, 8)
    sd[k] = stored
    sd["model.MinkUNet.final.kernel"] = sd["model.MinkUNet.final.kernel"].reshape(1, 8, 1)
    net.load_state_dict(sd)
    got = net.model.MinkUNet.state_dict()["block2.0.downsample.0.kernel"]
    assert tuple(got.shape) == (8, 16)
    np.testing.assert_array_equal(got.reshape(-1).numpy(), stored.reshape(-1).numpy())
    net.model.set_me_conventions(None)
    assert net.model.blob_permutation() is None
    # without the declared lin_layout a transposed 2-D kernel is a shape error, as in the reference's strict load
    with pytest.raises(RuntimeError):
        SPSNet(CFG).load_state_dict(sd)
