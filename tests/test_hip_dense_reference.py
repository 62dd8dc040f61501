"""The HIP path against a SECOND, independently written fp32 reference -- CustomMinkUNet14 built from torch's dense
F.conv3d / F.conv_transpose3d / F.batch_norm on a zero-filled grid (tests/dense_reference.py), run on the same GPU in fp32 --
compared DIRECTLY, not through `oracle/`.  It pins what the same-author oracle could share a mistake on:
the layer order and the operand order of the four concatenations (minkunet.py:161-219, :192, :200, :208, :216), the residual /
downsample wiring (resnet.py:98-108), the stride and transposed-convolution geometry and where BatchNorm / ReLU sit.  It does
not lift "parity unpinned" (DESIGN section 8): how a [K, C_in, C_out] tensor's K index maps to an offset is the repository's
reading of MinkowskiEngine, here as in the oracle.
Tolerances: logits <= 1e-3 (north_star), feature taps <= 5e-4, labels identical outside a 1e-5 band around the threshold."""
import numpy as np
import pytest
import torch

from sps_amd import synthetic
from tests.dense_reference import DenseSPS
from tests.helpers import CFG, net_from_params, straddle_params

pytestmark = pytest.mark.gpu

VS = CFG["MODEL"]["VOXEL_SIZE"]
EPS = CFG["FILTER"]["THRESHOLD"]
TAP = {"out_p1": ("conv0", 0), "block1": ("block1", 1), "block2": ("block2", 2), "block3": ("block3", 3), "block4": ("block4", 4),
       "block5": ("block5", 3), "block6": ("block6", 2), "block7": ("block7", 1), "block8": ("block8", 0)}


def _hip_tensors(c, name, level):
    """(coordinates [V,5], features [V,C]) of a tapped feature map of the last forward on context c, device tensors."""
    import ctypes as C
    from sps_amd import _native
    count = c.level_counts()[level]
    vox = torch.empty((count, 5), dtype=torch.int32, device="cuda")
    _native.check(_native.lib.sps_get_voxels(c.handle, level, vox.data_ptr()))
    r, k = C.c_int64(), C.c_int64()
    _native.check(_native.lib.sps_get_feature(c.handle, name.encode(), None, C.byref(r), C.byref(k)))
    feat = torch.empty((r.value, k.value), dtype=torch.float32, device="cuda")
    _native.check(_native.lib.sps_get_feature(c.handle, name.encode(), feat.data_ptr(), C.byref(r), C.byref(k)))
    assert r.value == count
    return vox, feat


def _random_weight_net(seed):
    """Reference-layout random parameters WITHOUT the oracle's generator: torch only (Kaiming-like kernels, BN statistics away
    from the identity); `final` rescaled on the dense network itself so that ~30 % of the scan scores exceed the threshold."""
    from sps_amd.models.models import SPSNet
    g = torch.Generator().manual_seed(seed)
    net = SPSNet(CFG)
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith(".kernel"):
            fan = v.shape[-1] * (v.shape[0] if v.dim() == 3 else 1)
            v.copy_(torch.randn(v.shape, generator=g) * (2.0 / fan) ** 0.5)
        elif k.endswith("bn.weight") or k.endswith("running_var"):
            v.copy_(torch.empty(v.shape).uniform_(0.5, 1.5, generator=g))
        elif k.endswith("bn.bias") or k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    return net, sd


@pytest.mark.parametrize("seed,shift", [(31, 0.0), (32, -7.3)])
def test_hip_path_equals_the_dense_torch_network(seed, shift):
    batch = synthetic.small_scene(seed=seed, n_scan=2500, extent=4.0)
    batch[:, 1:4] += shift                                           # (negative octants: floor strides)
    net, sd = _random_weight_net(seed)
    pts = torch.from_numpy(batch).cuda()
    # calibrate `final` on the dense network (no oracle in this test): ~30 % of the scan scores >= eps
    dense = DenseSPS(sd, device="cuda", dtype=torch.float32)
    _, lg = dense.forward(pts, VS)
    scan = pts[:, 4] == 1
    b0 = float(sd["model.MinkUNet.final.bias"].reshape(-1)[0])
    q = float(torch.quantile(8.0 * (lg[scan].double() - b0), 0.7))
    sd["model.MinkUNet.final.kernel"].mul_(8.0)
    sd["model.MinkUNet.final.bias"].fill_(float(np.log(EPS / (1 - EPS))) - q)
    net.load_state_dict(sd)
    net = net.cuda().eval().freeze()
    dense = DenseSPS(net.state_dict(), device="cuda", dtype=torch.float32)
    taps = {}
    ref_scores, ref_logits = dense.forward(pts, VS, taps=taps)
    got = net(pts)
    torch.cuda.synchronize()
    from sps_amd.models.models import get_context
    c = get_context(0)
    # --- the coordinate sets: every level's voxels are exactly the non-empty sites of the dense masks
    counts = c.level_counts()
    # --- every tapped feature map, row by row at the HIP path's own coordinates
    for name, (tap, level) in TAP.items():
        vox, feat = _hip_tensors(c, name, level)
        want = DenseSPS.rows_at(taps[tap], taps["_origin"], vox, 1 << level)
        assert want.shape == feat.shape, name
        err = float((feat - want).abs().max())
        assert err <= 5e-4, (name, err)
        active_sites = int((taps[tap].abs().sum(2) > 0).sum())
        assert active_sites <= counts[level], (name, active_sites, counts[level])     # nothing lives off the HIP path's coordinate set
    # --- logits (per point) and scores
    from sps_amd import _native
    logits_v = torch.empty(counts[0], dtype=torch.float32, device="cuda")
    _native.check(_native.lib.sps_get_logits(c.handle, logits_v.data_ptr()))
    inv = torch.empty(len(batch), dtype=torch.int64, device="cuda")
    _native.check(_native.lib.sps_get_inverse(c.handle, inv.data_ptr()))
    err_logit = float((logits_v[inv] - ref_logits).abs().max())
    assert err_logit <= 1e-3, err_logit
    assert float((got - ref_scores).abs().max()) <= 1e-4
    # --- labels: identical outside a 1e-5 band around eps, both classes present
    e = np.float32(EPS)
    s, r = got.cpu().numpy(), ref_scores.cpu().numpy()
    band = np.abs(r - e) > 1e-5
    np.testing.assert_array_equal((s < e)[band], (r < e)[band])
    frac = float((r[batch[:, 4] == 1] >= e).mean())
    assert 0.1 < frac < 0.6, frac
