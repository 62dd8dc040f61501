"""GPU parity of the baseline heads -- MOS4DNet / MapMOSNet on the HIP backbone against the
oracle's restatement (oracle/sps_oracle.py head_forward).  Per-point logits within 1e-3 (asserted tighter);
the labels (logit > 0: mos4d_node.py:114, mapmos.py:84-89) identical outside a tiny band around 0."""
import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from sps_amd import synthetic
from tests.helpers import net_from_params, state_dict_from_params

pytestmark = pytest.mark.gpu


def _buffer_cloud(n_scans, first_index, n_az=300, voxel=0.2):
    """A 4DMOS-style scan buffer: n_scans consecutive synthetic scans, t = running scan index."""
    rows = []
    for i in range(n_scans):
        s = synthetic.lidar_scan(seed=20 + i, x_offset=0.4 * i, n_beams=32, n_azimuth=n_az)
        t = np.full((len(s), 1), first_index + i, np.float32)
        rows.append(np.concatenate([np.zeros((len(s), 1), np.float32), s[:, :3], t], 1))
    return np.concatenate(rows, 0).astype(np.float32)


def _labels_agree(got, want, band=1e-5):
    keep = np.abs(want) > band
    np.testing.assert_array_equal((got > 0)[keep], (want > 0)[keep])


@pytest.fixture(scope="module")
def mos4d():
    from sps_amd.models.baselines import MOS4DNet
    p = O.random_params(seed=4, out_channels=3)
    m = MOS4DNet(0.2)
    # the node loads {k.replace("model.MinkUNet.", ""): v} into model.MinkUNet (mos4d_node.py:63-70)
    m.MinkUNet.load_state_dict(state_dict_from_params(p, prefix=""))
    return p, m.cuda().eval().freeze()


@pytest.fixture(scope="module")
def mapmos():
    from sps_amd.models.baselines import MapMOSNet
    p = O.random_params(seed=6, out_channels=1)
    m = MapMOSNet(0.1)
    m.MinkUNet.load_state_dict(state_dict_from_params(p, prefix=""))
    return p, m.cuda().eval().freeze()


@pytest.mark.parametrize("first_index", [0, 1234])
def test_mos4d_ten_scan_buffer(mos4d, first_index):
    p, m = mos4d
    c = _buffer_cloud(10, first_index)
    got = m(torch.from_numpy(c).cuda())
    torch.cuda.synchronize()
    want, _ = O.head_forward(p, c, 0.2)
    assert got.shape == (len(c),)
    g = got.cpu().numpy()
    np.testing.assert_allclose(g, want[:, 2], rtol=0, atol=2e-4)
    _labels_agree(g, want[:, 2])
    from sps_amd.models.models import get_context
    get_context(0).check_errors(torch.cuda.current_stream().cuda_stream)   # the re-based t fits the key range


def test_mos4d_all_columns_through_c_abi(mos4d):
    """sps_forward_head itself: all 3 columns, padded row stride, sigmoid activation."""
    p, m = mos4d
    c = _buffer_cloud(4, 3, n_az=200)
    dev = torch.from_numpy(c).cuda()
    from sps_amd.models.models import get_context
    st = torch.cuda.current_stream().cuda_stream
    ctx = get_context(0, st)
    m._sync_weights(ctx)
    out = torch.full((len(c), 5), -7.0, device="cuda")
    ctx.forward_head(dev.data_ptr(), 5, len(c), 0.2, None, 0.0, out.data_ptr(), 5, 1, st)
    torch.cuda.synchronize()
    want, _ = O.head_forward(p, c, 0.2)
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[:, :3], O.sigmoid(want), rtol=0, atol=1e-4)
    assert (o[:, 3:] == -7.0).all()                       # columns past out_channels are left alone


def test_mapmos_predict(mapmos):
    p, m = mapmos
    scan = synthetic.lidar_scan(seed=31, n_beams=32, n_azimuth=500)[:, :3]
    mp = synthetic.build_map(offsets=(-2.0, 0.0, 2.0), n_beams=32, n_azimuth=500)[:, :3]
    scan_t, map_t = torch.from_numpy(scan).cuda(), torch.from_numpy(mp).cuda()
    si = torch.ones(len(scan), 1, device="cuda")           # mapmos_node.py:92-93
    mi = torch.zeros(len(mp), 1, device="cuda")
    ls, lm = m.predict(scan_t, map_t, si, mi)
    torch.cuda.synchronize()
    assert ls.shape == (len(scan),) and lm.shape == (len(mp),)
    c = np.concatenate([np.concatenate([np.zeros((len(scan), 1), np.float32), scan, np.zeros((len(scan), 1), np.float32)], 1),
                        np.concatenate([np.zeros((len(mp), 1), np.float32), mp, -np.ones((len(mp), 1), np.float32)], 1)], 0)
    idx = np.concatenate([np.ones(len(scan), np.float32), np.zeros(len(mp), np.float32)])
    want, info = O.head_forward(p, c, 0.1, features=O.mapmos_features(idx))
    assert set(np.unique(info["voxel_features"])) == {1.0, 2.0}
    got = np.concatenate([ls.cpu().numpy(), lm.cpu().numpy()])
    np.testing.assert_allclose(got, want[:, 0], rtol=0, atol=2e-4)
    _labels_agree(got, want[:, 0])
    lab = m.to_label(ls)
    np.testing.assert_array_equal(lab.cpu().numpy(), (ls.cpu().numpy() > 0).astype(np.float32))


def test_mapmos_mixed_indices_voxel_mean(mapmos):
    """Several indices inside one voxel: the conv0 input is the per-voxel MEAN of the point features."""
    p, m = mapmos
    rng = np.random.default_rng(3)
    n = 6000
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    xyz[:, 2] = rng.uniform(-0.1, 0.1, n)
    t = rng.integers(-1, 1, n).astype(np.float32)
    c = np.concatenate([np.zeros((n, 1), np.float32), xyz, t[:, None]], 1).astype(np.float32)
    idx = rng.integers(0, 7, n).astype(np.float32)
    got = m(torch.from_numpy(c).cuda(), torch.from_numpy(idx).cuda().reshape(-1, 1))
    torch.cuda.synchronize()
    want, info = O.head_forward(p, c, 0.1, features=O.mapmos_features(idx))
    assert len(info["voxels"]) < 0.8 * n                     # voxels really hold several points
    np.testing.assert_allclose(got.cpu().numpy(), want[:, 0], rtol=0, atol=2e-4)
    again = m(torch.from_numpy(c).cuda(), torch.from_numpy(idx).cuda().reshape(-1, 1))
    torch.cuda.synchronize()
    assert torch.equal(got, again)                           # fixed-point feature sums: bit-reproducible


def test_head_and_sps_models_share_a_context(mos4d):
    """The native context of a (device, stream) is shared: switching between a 3-channel head and the SPS
    model re-uploads the right blob, and sps_forward refuses a context that holds a k-channel head."""
    p3, m = mos4d
    sp = O.random_params(seed=0)
    net = net_from_params(sp).cuda().eval().freeze()
    batch = synthetic.small_scene(seed=2, n_scan=1500)
    dev = torch.from_numpy(batch).cuda()
    s1 = net(dev).cpu().numpy()
    c = _buffer_cloud(3, 0, n_az=150)
    g = m(torch.from_numpy(c).cuda()).cpu().numpy()
    s2 = net(dev).cpu().numpy()
    np.testing.assert_array_equal(s1, s2)
    want, _ = O.head_forward(p3, c, 0.2)
    np.testing.assert_allclose(g, want[:, 2], rtol=0, atol=2e-4)
    from sps_amd import _native
    from sps_amd.models.models import get_context
    st = torch.cuda.current_stream().cuda_stream
    ctx = get_context(0, st)
    m._sync_weights(ctx)
    out = torch.empty(len(batch), device="cuda")
    with pytest.raises(_native.SpsError, match="3-channel head"):
        ctx.forward(dev.data_ptr(), dev.stride(0), len(batch), 0.1, out.data_ptr(), st)


def test_head_edge_cases(mos4d, mapmos):
    _, m4 = mos4d
    _, mm = mapmos
    assert m4(torch.empty((0, 5), device="cuda")).shape == (0,)
    one = torch.tensor([[0.0, 0.3, -0.2, 0.1, 77.0]], device="cuda")
    assert torch.isfinite(m4(one)).all()
    with pytest.raises(RuntimeError, match="MI355X"):
        m4(torch.zeros((4, 5)))
    with pytest.raises(ValueError, match="features for"):
        mm._run(torch.zeros((4, 5), device="cuda"), torch.ones(3, device="cuda"), 0.1)
    # a buffer spanning more than 32 time slices cannot be keyed: reported, not silently wrong
    c = torch.zeros((40, 5), device="cuda")
    c[:, 4] = torch.arange(40, device="cuda")
    c[:, 1] = torch.arange(40, device="cuda") * 0.5
    out = m4(c)
    from sps_amd import _native
    from sps_amd.models.models import get_context
    with pytest.raises(_native.SpsError):
        get_context(0).check_errors(torch.cuda.current_stream().cuda_stream)
    assert torch.isnan(out).any()
