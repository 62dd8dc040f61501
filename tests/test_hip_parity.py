"""GPU parity tests: the HIP path (through the C ABI / the sps.* drop-in API) against the oracle
on the same seeded inputs.  Integer work (voxels, inverse map, parents, kernel-map pair counts,
labels, confusion counts) must be exact; logits within 1e-3 (north_star), asserted tighter."""
import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from sps_amd import synthetic
from tests.helpers import CFG, assert_nondegenerate, net_from_params, plant_threshold_labels, straddle_params

pytestmark = pytest.mark.gpu

VS = CFG["MODEL"]["VOXEL_SIZE"]
EPS = CFG["FILTER"]["THRESHOLD"]


@pytest.fixture(scope="module")
def params():
    """Seed-0 synthetic weights with `final` rescaled so that ~30 % of the scan scores are >= eps = 0.84 (25-30 % on
    every scene used below): labels, TP / FP / FN / TN and dIoU are compared in the regime where both classes occur."""
    return straddle_params(O.random_params(seed=0), synthetic.small_scene(seed=11, n_scan=2500))


@pytest.fixture(scope="module")
def net(params):
    assert torch.cuda.is_available()
    return net_from_params(params).cuda().eval().freeze()


def ctx():
    from sps_amd.models.models import get_context
    return get_context(0)


def run(net, batch_np):
    dev = torch.from_numpy(np.ascontiguousarray(batch_np)).cuda()
    scores = net(dev)
    torch.cuda.synchronize()
    return dev, scores


def get_voxels_of(c, level, count):
    out = torch.empty((count, 5), dtype=torch.int32, device="cuda")
    from sps_amd import _native
    _native.check(_native.lib.sps_get_voxels(c.handle, level, out.data_ptr()))
    return out.cpu().numpy()


def get_voxels(level, count):
    return get_voxels_of(ctx(), level, count)


def get_feature(name):
    import ctypes as C
    from sps_amd import _native
    r, c = C.c_int64(), C.c_int64()
    _native.check(_native.lib.sps_get_feature(ctx().handle, name.encode(), None, C.byref(r), C.byref(c)))
    out = torch.empty((r.value, c.value), dtype=torch.float32, device="cuda")
    _native.check(_native.lib.sps_get_feature(ctx().handle, name.encode(), out.data_ptr(), C.byref(r), C.byref(c)))
    return out.cpu().numpy()


def match_rows(got: np.ndarray, want: np.ndarray) -> np.ndarray:
    """perm with got[i] == want[perm[i]]; asserts that the two coordinate sets are identical.
    (The HIP path orders voxel rows block by block, the oracle by first occurrence: row order is
    free, App. A.3 -- only sets, inverse maps and per-row values are compared.)"""
    assert got.shape == want.shape
    if len(got) == 0:
        return np.zeros(0, np.int64)
    og = np.lexsort(got.T[::-1])
    ow = np.lexsort(want.T[::-1])
    np.testing.assert_array_equal(got[og], want[ow])
    assert len(np.unique(got, axis=0)) == len(got), "duplicate voxel rows"
    perm = np.empty(len(got), np.int64)
    perm[og] = ow
    return perm


def oracle_table(kmap, n_out):
    """ME-style kernel map (per offset the (in, out) index lists) -> dense [K, n_out] table of input rows, -1 = no pair;
    asserts that an output row occurs at most once per offset."""
    tab = np.full((len(kmap), n_out), -1, np.int64)
    for k, (i, o) in enumerate(kmap):
        assert len(np.unique(o)) == len(o)
        tab[k, o] = i
    return tab


def assert_same_pairs(got, want, perm_out, perm_in, what):
    """``got`` [K, V_out] in the HIP path's row numbering == ``want`` in the oracle's: per offset the SET of (in, out) pairs."""
    got = np.asarray(got, np.int64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert got.min(initial=-1) >= -1 and got.max(initial=-1) < len(perm_in), what
    mapped = np.where(got >= 0, perm_in[np.maximum(got, 0)], -1)
    np.testing.assert_array_equal(mapped, want[:, perm_out], err_msg=what)


def check_kernel_maps(cm, perm, counts):
    """Row a10 as pair SETS: the 3x3x3x3 map of every level from the neighbour table AND (levels that run pair-exact) decoded
    from the rulebook the convolutions read, the 5x5x5x1 map, and the four stride-2 maps (= the transposed convolutions' maps
    with in / out swapped) against cm.k3 / k5 / kdown of the oracle."""
    c = ctx()
    n_pairs = 0
    for l in range(5):
        want = oracle_table(cm.k3(1 << l), counts[l])
        assert_same_pairs(c.kernel_map(l, 0).cpu().numpy(), want, perm[l], perm[l], f"3^4 neighbour table, level {l}")
        if l <= 1:
            tab, n_entries = c.kernel_map(l, 1)
            assert n_entries == int((want >= 0).sum()), f"rulebook level {l}: {n_entries} pairs, oracle {(want >= 0).sum()}"
            assert_same_pairs(tab.cpu().numpy(), want, perm[l], perm[l], f"3^4 rulebook, level {l}")
        n_pairs += int((want >= 0).sum())
    assert_same_pairs(c.kernel_map(5, 0).cpu().numpy(), oracle_table(cm.k5(), counts[0]), perm[0], perm[0], "5x5x5x1 map")
    for l in range(1, 5):
        want = oracle_table(cm.kdown(1 << (l - 1)), counts[l])
        assert int((want >= 0).sum()) == counts[l - 1]             # every fine voxel has exactly one parent and one offset
        assert_same_pairs(c.kernel_map(5 + l, 0).cpu().numpy(), want, perm[l], perm[l - 1], f"stride map into level {l}")
    return n_pairs


def check_full(net, params, batch, tol=2e-4, both_classes=False):
    dev, scores = run(net, batch)
    ref, info = O.sps_forward(params, batch[:, :5], VS, keep=True)
    counts = ctx().level_counts()
    cm = info["cm"]
    # --- integer structure: exact (as sets; rows matched by coordinate)
    perm = []
    for l in range(5):
        ts = 1 << l
        assert counts[l] == len(cm.coords[ts]), f"level {l}"
        perm.append(match_rows(get_voxels(l, counts[l]), cm.coords[ts]))
    from sps_amd import _native
    inv = torch.empty(len(batch), dtype=torch.int64, device="cuda")
    _native.check(_native.lib.sps_get_inverse(ctx().handle, inv.data_ptr()))
    np.testing.assert_array_equal(perm[0][inv.cpu().numpy()], info["inverse"])
    for l in range(4):
        par = torch.empty(counts[l], dtype=torch.int32, device="cuda")
        _native.check(_native.lib.sps_get_parent(ctx().handle, l, par.data_ptr()))
        np.testing.assert_array_equal(perm[l + 1][par.cpu().numpy()], cm.parent[1 << l][perm[l]])
    for l in range(5):
        want = [len(i) for i, _ in cm.k3(1 << l)]
        assert ctx().map_pairs(l) == want, f"3^4 map level {l}"
    assert ctx().map_pairs(5) == [len(i) for i, _ in cm.k5()]
    check_kernel_maps(cm, perm, counts)                           # ... and as pair sets, offset by offset
    # --- features
    tap_level = {"out_p1": 0, "block1": 1, "block2": 2, "block3": 3, "block4": 4, "block5": 3, "block6": 2,
                 "block7": 1, "block8": 0}
    for name, want in info["inter"].items():
        got = get_feature(name)
        np.testing.assert_allclose(got, want[perm[tap_level[name]]], rtol=tol, atol=tol, err_msg=name)
    logits = torch.empty(counts[0], dtype=torch.float32, device="cuda")
    _native.check(_native.lib.sps_get_logits(ctx().handle, logits.data_ptr()))
    np.testing.assert_allclose(logits.cpu().numpy(), info["logits"][perm[0]], rtol=0, atol=1e-3)
    s = scores.cpu().numpy()
    np.testing.assert_allclose(s, ref, rtol=0, atol=1e-4)
    # --- labels: identical outside a tiny band around the threshold
    e = np.float32(EPS)
    band = np.abs(ref - e) > 1e-5
    np.testing.assert_array_equal((s < e)[band], (ref < e)[band])
    if both_classes:
        # not a comparison of all-stable with all-stable: both labels occur, on both sides
        assert 0.05 < (ref[band] >= e).mean() < 0.95, "degenerate label distribution"
    return dev, s, ref


def test_small_scene_full_parity(net, params):
    check_full(net, params, synthetic.small_scene(seed=0, n_scan=2000), both_classes=True)


def test_launch_geometry_hint_does_not_change_a_bit(net, params):
    """sps_ctx_set_pipelined: one column tile per wave at the coarse levels (a context whose forwards run one after another,
    the default) or two (several contexts in flight: ScanEngine) -- the per-element summation order is the same, so scores
    and every tapped feature map are bit-identical; the engine's pipelines agree with a direct call for the same reason."""
    batch = synthetic.make_scene(scan_seed=3, n_azimuth=500)["batch"]        # LiDAR-like: levels 3 and 4 hold few tiles
    out = {}
    try:
        for mode in (False, True):
            ctx().set_pipelined(mode)
            _, scores = run(net, batch)
            out[mode] = (scores.clone(), {n: get_feature(n) for n in ("block3", "block4", "block5", "block8")})
    finally:
        ctx().set_pipelined(False)
    assert torch.equal(out[False][0], out[True][0])
    for n, f in out[False][1].items():
        np.testing.assert_array_equal(f, out[True][1][n], err_msg=n)
    assert float(out[False][0].min()) >= 0 and not torch.isnan(out[False][0]).any()


def test_other_seed_and_default_bn(params):
    p = O.random_params(seed=7, randomize_bn=False)
    n = net_from_params(p).cuda().eval().freeze()
    check_full(n, p, synthetic.small_scene(seed=5, n_scan=1500))


def test_negative_octants_and_requantised_corners(net, params):
    """Submap rows are voxel corners ix*0.1f that re-quantise to ix-1 for some negatives (App. E)."""
    b = synthetic.small_scene(seed=2, n_scan=1200, extent=4.0)
    b[:, 1:4] -= 7.3
    check_full(net, params, b, both_classes=True)


def oracle_confusion(scores, batch):
    """[TP, FP, FN, TN] over the scan rows with the reference's rule (models.py:97-98: `<` on float32)."""
    scan = batch[:, 4] == 1
    e = np.float32(EPS)
    pred = scores[scan].astype(np.float32) >= e
    gt = batch[scan, 5].astype(np.float32) >= e
    return np.array([(gt & pred).sum(), (~gt & pred).sum(), (gt & ~pred).sum(), (~gt & ~pred).sum()], np.float64)


def test_metrics_match_oracle(net, params):
    """predict_step vs the oracle in the regime where pred == 1 and gt == 1 both occur (TP, FP, FN, TN > 0), with
    labels planted exactly at / one ulp around eps."""
    batch = plant_threshold_labels(synthetic.small_scene(seed=11, n_scan=2500))
    dev, s, ref = check_full(net, params, batch, both_classes=True)
    m = net.predict_step(dev, 0)
    assert_nondegenerate([m["count"], m["tp"], m["fp"], m["fn"], m["tn"], 0, 0, 0])
    assert min(m["precision"], m["recall"], m["f1"], m["dIoU"]) > 0
    np.testing.assert_array_equal([m["tp"], m["fp"], m["fn"], m["tn"]], oracle_confusion(s, batch))
    mo = O.predict_metrics(s, batch, EPS)          # same scores -> counts must be exact
    for k in ("precision", "recall", "f1", "accuracy", "dIoU"):
        assert m[k] == pytest.approx(mo[k], abs=1e-12), k
    assert m["loss"] == pytest.approx(mo["loss"], rel=1e-9)
    assert m["r2"] == pytest.approx(mo["r2"], rel=1e-8)
    assert net.dIoU[-1] == m["dIoU"] and len(net.predict_loss) >= 1
    # against the ORACLE's scores: the confusion counts may differ only by rows whose score lies within 1e-5 of eps
    n_band = int((np.abs(ref[batch[:, 4] == 1] - np.float32(EPS)) <= 1e-5).sum())
    assert np.abs(np.array([m["tp"], m["fp"], m["fn"], m["tn"]]) - oracle_confusion(ref, batch)).max() <= n_band
    mr = O.predict_metrics(ref, batch, EPS)
    assert m["dIoU"] == pytest.approx(mr["dIoU"], abs=(n_band + 1e-9) / max(m["tp"] + m["fp"] + m["fn"], 1))
    assert m["loss"] == pytest.approx(mr["loss"], abs=1e-6)


def test_threshold_ties_scores_and_labels(net):
    """sps_metrics on hand-made scores: score == eps is class 1 (`score < eps ? 0 : 1` in float32, models.py:97), the
    float32 below it class 0; the same for labels; eps itself is the float32 rounding of the config's 0.84."""
    e = np.float32(EPS)
    lo, hi = np.nextafter(e, np.float32(0)), np.nextafter(e, np.float32(1))
    vals = np.array([e, lo, hi, 0.0, 1.0, 0.5], np.float32)
    sc, lb = np.meshgrid(vals, vals, indexing="ij")
    n = sc.size
    batch = np.zeros((n + 3, 6), np.float32)
    batch[:n, 4] = 1
    batch[:n, 5] = lb.ravel()
    batch[n:, 5] = 1.0                                  # map rows (t = 0) never count
    scores = np.concatenate([sc.ravel(), np.ones(3, np.float32)])
    dev, sd = torch.from_numpy(batch).cuda(), torch.from_numpy(scores).cuda()
    got = np.asarray(net.step_metrics(dev, sd, 1), np.float64)[0]
    assert got[0] == n
    np.testing.assert_array_equal(got[1:5], oracle_confusion(scores, batch))
    assert_nondegenerate(got)
    pred = np.where(sc.ravel() < e, 0, 1)
    assert pred[:6].tolist() == [1] * 6 and pred[6:12].tolist() == [0] * 6      # score == eps -> 1, just below -> 0
    want = O.predict_metrics(scores, batch, EPS)
    from sps_amd.models.models import metrics_from_sums
    mg = metrics_from_sums(got)
    for k in ("precision", "recall", "f1", "accuracy", "dIoU"):
        assert mg[k] == pytest.approx(want[k], abs=1e-12), k


def test_batch_independence(net, params):
    """b = 0..2 batched == three single runs (b is never convolved across, App. A.5)."""
    parts = [synthetic.small_scene(seed=20 + i, n_scan=900) for i in range(3)]
    singles = []
    for p in parts:
        _, s = run(net, p)
        singles.append(s.cpu().numpy())
    stacked = []
    for i, p in enumerate(parts):
        q = p.copy()
        q[:, 0] = i
        stacked.append(q)
    big = np.concatenate(stacked, 0)
    dev, s = run(net, big)
    # same terms per row; only the f32 summation grouping may differ with the tile composition
    np.testing.assert_allclose(s.cpu().numpy(), np.concatenate(singles), rtol=0, atol=2e-6)
    per = net.step_metrics(dev, s, n_batches=3)
    for i in range(3):
        assert_nondegenerate(per[i])
    for i, p in enumerate(parts):
        sb = s.cpu().numpy()[sum(len(q) for q in parts[:i]): sum(len(q) for q in parts[:i + 1])]
        mo = O.predict_metrics(sb, p, EPS)
        from sps_amd.models.models import metrics_from_sums
        assert metrics_from_sums(per[i])["dIoU"] == pytest.approx(mo["dIoU"], abs=1e-12, nan_ok=True)


def test_permutation_and_duplicates(net, params):
    batch = synthetic.small_scene(seed=4, n_scan=1000)
    _, s0 = run(net, batch)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(batch))
    _, s1 = run(net, batch[perm])
    np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy()[perm], rtol=0, atol=2e-6)   # order-free
    dup = np.concatenate([batch, batch[:300]], 0)
    _, s2 = run(net, dup)
    np.testing.assert_array_equal(s2.cpu().numpy()[: len(batch)], s0.cpu().numpy())     # same voxel rows
    np.testing.assert_array_equal(s2.cpu().numpy()[len(batch):], s0.cpu().numpy()[:300])


def test_run_to_run_determinism(net, params):
    """No atomics in any floating-point sum: the same input gives the same bits, every time."""
    batch = synthetic.small_scene(seed=31, n_scan=3000)
    _, a = run(net, batch)
    a = a.cpu().numpy().copy()
    for _ in range(3):
        _, b = run(net, batch)
        np.testing.assert_array_equal(b.cpu().numpy(), a)


def test_edge_cases(net, params):
    # empty
    e = torch.empty((0, 6), dtype=torch.float32, device="cuda")
    assert net(e).shape == (0,)
    # single point: conv chain on one voxel
    one = np.array([[0, 1.23, -4.56, 0.78, 1, 0.3]], np.float32)
    check_full(net, params, one)
    # two points in one voxel get the same score
    two = np.array([[0, 1.21, -4.56, 0.78, 1, 0.3], [0, 1.29, -4.51, 0.71, 1, 0.9]], np.float32)
    _, s = run(net, two)
    assert s[0].item() == s[1].item()


def test_out_of_range_is_reported(net):
    from sps_amd._native import SpsError
    bad = np.array([[0, 2.0e4, 0, 0, 1, 0.5], [0, 1.0, 1.0, 1.0, 1, 0.5]], np.float32)   # 200 000 voxels away
    dev = torch.from_numpy(bad).cuda()
    s = net(dev)
    with pytest.raises(SpsError) as ei:
        ctx().check_errors(torch.cuda.current_stream().cuda_stream)
    assert ei.value.code == -4
    s = s.cpu().numpy()
    assert np.isnan(s[0]) and np.isfinite(s[1])
    ctx().check_errors(torch.cuda.current_stream().cuda_stream)          # flag was cleared


def test_quantisation_probe_matches_f32_division():
    """SURVEY App. C probe: floor(x / f32(0.1)) must be IEEE f32 division (not x*10, not f64)."""
    rng = np.random.default_rng(123)
    n = 1_000_000
    pts = np.zeros((n, 5), np.float32)
    pts[:, 1:4] = rng.uniform(-100, 100, (n, 3)).astype(np.float32)
    pts[:, 4] = 1
    p = O.random_params(seed=1)
    net = net_from_params(p).cuda().eval().freeze()
    dev = torch.from_numpy(pts).cuda()
    net(dev)
    counts = ctx().level_counts()
    vox = get_voxels(0, counts[0])
    want, inv = O.unique_first(O.quantize(pts, VS))
    match_rows(vox, want)


def test_prune_matches_oracle():
    import sps.datasets.util as util
    rng = np.random.default_rng(5)
    map_xyz = rng.uniform(-8, 8, (20000, 3)).astype(np.float32)
    scan_xyz = (map_xyz[:6000] + rng.normal(0, 0.05, (6000, 3))).astype(np.float32)
    scan_xyz = np.concatenate([scan_xyz, scan_xyz[:500]], 0)                     # duplicates
    mcf = util.to_coords_features(torch.from_numpy(map_xyz).cuda(), "map", VS)
    scf = util.to_coords_features(torch.from_numpy(scan_xyz).cuda(), "scan", VS)
    np.testing.assert_array_equal(scf.cloud_coords.cpu().numpy(), O.to_coords(scan_xyz, VS))
    sub, n_scan_vox = util.prune(mcf, scf, VS)
    want, n_want = O.prune(O.to_coords(map_xyz, VS), O.to_coords(scan_xyz, VS), VS)
    assert n_scan_vox == n_want
    np.testing.assert_array_equal(sub.cpu().numpy(), want)       # same order: scan first-occurrence
    # float entry point (fused truncation) gives the same rows
    from sps_amd.models.models import get_context
    c = get_context(0)
    st = torch.cuda.current_stream().cuda_stream
    m = torch.from_numpy(map_xyz).cuda(); sc = torch.from_numpy(scan_xyz).cuda()
    c.map_upload(m.data_ptr(), 3, len(m), VS, st)
    out = torch.empty((len(sc), 3), dtype=torch.float32, device="cuda")
    a, b = c.submap_voxel(sc.data_ptr(), 3, len(sc), out.data_ptr(), st)
    assert (a, b) == (len(want), n_want)
    np.testing.assert_array_equal(out[:a].cpu().numpy(), want)
    util._MAP_CACHE.clear()


def test_infer_helper(net, params):
    import sps.datasets.util as util
    batch = synthetic.small_scene(seed=9, n_scan=800)
    ns = int((batch[:, 4] == 1).sum())
    scan = torch.from_numpy(batch[:ns, 1:4]).cuda()
    sub = torch.from_numpy(batch[ns:, 1:4]).cuda()
    scores, dt = util.infer(scan, sub, net)
    ref, _ = O.sps_forward(params, batch[:, :5], VS)
    np.testing.assert_allclose(scores.cpu().numpy(), ref[:ns], atol=1e-4)
    with pytest.raises(AssertionError, match="Expected 3 columns"):
        util.infer(torch.zeros(4, 4).cuda(), sub, net)


@pytest.mark.timeout(600)
def test_config2_full_size_parity(net, params):
    """BASELINE config 2 (~100k-pt scan + submap, 0.1 m): full oracle comparison."""
    sc = synthetic.make_scene(scan_seed=1)
    check_full(net, params, sc["batch"], tol=5e-4, both_classes=True)


def test_lightning_checkpoint_roundtrip(tmp_path, params):
    """predict.py:56-58 / util.py:29-46: a Lightning-style .ckpt ({"state_dict": {"model.MinkUNet.*"}}) loads through
    util.load_model (key renaming, MOSLoss entries dropped) and through SPSNet.load_state_dict."""
    import sps.datasets.util as util
    from tests.helpers import state_dict_from_params
    sd = state_dict_from_params(params)
    sd["model.MOSLoss.weight"] = torch.zeros(3)                       # dropped by load_model
    path = str(tmp_path / "420_601.ckpt")
    torch.save({"state_dict": sd, "hyper_parameters": CFG}, path)
    model = util.load_model(CFG, path)
    assert not any(p.requires_grad for p in model.parameters()) and not model.training
    batch = synthetic.small_scene(seed=41, n_scan=1500)
    ref, _ = O.sps_forward(params, batch[:, :5], VS)
    s = model(torch.from_numpy(batch).cuda()).cpu().numpy()
    np.testing.assert_allclose(s, ref, rtol=0, atol=1e-4)


def test_config3_batch4_streamed(net, params):
    """BASELINE config 3: batch = 4 scans in one tensor (collate layout), per-scan metric rows."""
    from sps.datasets.blt_dataset import BacchusModule
    from sps_amd.models.models import metrics_from_sums
    items = [torch.from_numpy(plant_threshold_labels(synthetic.small_scene(seed=60 + i, n_scan=2500))[:, 1:]) for i in range(4)]
    batch = BacchusModule.collate_fn(items)                       # [sum N, 6] with b = 0..3
    dev = batch.cuda()
    s = net(dev)
    per = net.step_metrics(dev, s, n_batches=4)
    sc = s.cpu().numpy()
    b = batch.numpy()
    ref, _ = O.sps_forward(params, b[:, :5], VS)
    np.testing.assert_allclose(sc, ref, rtol=0, atol=1e-4)
    for i in range(4):
        rows = b[:, 0] == i
        assert_nondegenerate(per[i])
        np.testing.assert_array_equal(per[i][1:5], oracle_confusion(sc[rows], b[rows]))
        mo = O.predict_metrics(sc[rows], b[rows], EPS)
        mg = metrics_from_sums(per[i])
        for k in ("precision", "recall", "f1", "accuracy", "dIoU"):
            assert mg[k] == pytest.approx(mo[k], abs=1e-12, nan_ok=True), (i, k)
        assert mg["loss"] == pytest.approx(mo["loss"], rel=1e-9)


@pytest.mark.timeout(900)
def test_config3_batch4_full_size(net, params):
    """BASELINE config 3 at spec size: four consecutive ~150k-row scans in one forward (collate_fn layout, ~600k rows)
    against the C oracle: scores, labels, per-scan confusion counts; and batch independence at full size (every scan's
    scores equal its single-scan forward to the rounding of a different tile composition)."""
    from oracle import c_oracle
    from sps_amd.models.models import metrics_from_sums
    scans = [plant_threshold_labels(b) for b in synthetic.make_sequence(4)]
    batch = synthetic.collate(scans)
    assert len(batch) > 550_000 and sorted(np.unique(batch[:, 0])) == [0, 1, 2, 3]
    dev = torch.from_numpy(batch).cuda()
    s, sums = net.forward_metrics(dev, 4)
    torch.cuda.synchronize()
    sg, sums = s.cpu().numpy(), sums.cpu().numpy()
    ref, info = c_oracle.forward(c_oracle.pack_blob(params), batch[:, :5], VS, nthreads=8)
    assert ctx().level_counts() == info["level_counts"]
    np.testing.assert_allclose(sg, ref, rtol=0, atol=1e-4)
    e = np.float32(EPS)
    band = np.abs(ref - e) > 1e-5
    np.testing.assert_array_equal((sg < e)[band], (ref < e)[band])
    off = 0
    for b, scan in enumerate(scans):
        rows = slice(off, off + len(scan))
        off += len(scan)
        assert_nondegenerate(sums[b])
        np.testing.assert_array_equal(sums[b, 1:5], oracle_confusion(sg[rows], batch[rows]))
        n_band = int((~band[rows] & (batch[rows, 4] == 1)).sum())
        assert np.abs(sums[b, 1:5] - oracle_confusion(ref[rows], batch[rows])).max() <= n_band
        single = net(torch.from_numpy(scan).cuda()).cpu().numpy()
        np.testing.assert_allclose(sg[rows], single, rtol=0, atol=5e-6)
        assert metrics_from_sums(sums[b])["dIoU"] > 0


@pytest.mark.timeout(900)
def test_config2_kernel_maps_as_pair_sets(net):
    """BASELINE config 2 at spec size: all ten kernel maps (+ the rulebooks of the pair-exact levels) as sets of (in, out)
    pairs per offset against the numpy oracle's coordinate manager; also from an inference-only context (the product
    loop's kind: ScanEngine), where the rulebook is all that exists at levels 0-1."""
    batch = synthetic.make_scene(scan_seed=1)["batch"]
    assert len(batch) > 140_000
    run(net, batch)
    counts = ctx().level_counts()
    vox, inv = O.unique_first(O.quantize(batch[:, :5], VS))
    cm = O.CoordinateManager(vox)
    for ts in (2, 4, 8, 16):
        cm.ensure_stride(ts)
    perm = [match_rows(get_voxels(l, counts[l]), cm.coords[1 << l]) for l in range(5)]
    n_pairs = check_kernel_maps(cm, perm, counts)
    assert n_pairs > 2_500_000                                    # (3^4 maps alone; + 2.7 M pairs of the 5x5x5x1 map)
    from sps_amd import _native
    from sps_amd.models.models import get_context
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        c2 = get_context(0, st.cuda_stream)
        c2.set_inference_only(True)
        try:
            dev = torch.from_numpy(batch).cuda()
            net(dev)
            st.synchronize()
            assert c2.level_counts() == counts
            with pytest.raises(_native.SpsError):
                c2.kernel_map(0, 0)                                # no neighbour table at the pair-exact levels
            perm2 = [match_rows(get_voxels_of(c2, l, counts[l]), cm.coords[1 << l]) for l in range(2)]
            for l in range(2):
                tab, n_entries = c2.kernel_map(l, 1)
                want = oracle_table(cm.k3(1 << l), counts[l])
                assert n_entries == int((want >= 0).sum())
                assert_same_pairs(tab.cpu().numpy(), want, perm2[l], perm2[l], f"inference-only rulebook, level {l}")
        finally:
            c2.set_inference_only(False)


@pytest.mark.timeout(900)
def test_config4_nclt_size_properties(net, params):
    """BASELINE config 4 at spec size (NCLT-like: three merged 128-beam scans to 100 m against a 25-position map ->
    ~520k rows, >= 300k active level-0 voxels): too big for a per-feature diff to be cheap, so check the scores against
    the C oracle plus size-independent structural invariants."""
    from oracle import c_oracle
    sc = synthetic.make_nclt_scene(seed=5)
    batch = sc["batch"]
    assert len(batch) > 500_000
    dev, s = run(net, batch)
    counts = ctx().level_counts()
    assert counts[0] >= 300_000 and all(counts[i] > counts[i + 1] > 0 for i in range(4))
    blob = c_oracle.pack_blob(params)
    ref, info = c_oracle.forward(blob, batch[:, :5], VS, nthreads=8)
    assert counts == info["level_counts"]
    sg = s.cpu().numpy()
    np.testing.assert_allclose(sg, ref, rtol=0, atol=1e-4)
    e = np.float32(EPS)
    band = np.abs(ref - e) > 1e-5
    np.testing.assert_array_equal((sg < e)[band], (ref < e)[band])
    assert 0.05 < (ref[band] >= e).mean() < 0.95, "degenerate label distribution"
    sums = np.asarray(net.step_metrics(dev, s, 1), np.float64)[0]
    assert_nondegenerate(sums)
    np.testing.assert_array_equal(sums[1:5], oracle_confusion(sg, batch))
    # points of one voxel share one score (App. A.15): scores are a function of the inverse map
    from sps_amd import _native
    inv = torch.empty(len(batch), dtype=torch.int64, device="cuda")
    _native.check(_native.lib.sps_get_inverse(ctx().handle, inv.data_ptr()))
    inv = inv.cpu().numpy()
    first = np.full(counts[0], -1.0, np.float32)
    first[inv] = sg
    np.testing.assert_array_equal(first[inv], sg)
    # 3^4 kernel map symmetry: pairs(k) == pairs(80 - k)
    p = ctx().map_pairs(0)
    assert p == p[::-1] and p[40] == counts[0]


def test_predict_cli_synthetic():
    """scripts/predict.py end to end on two synthetic scans: prints the reference's six metric lines."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "predict.py"), "--synthetic", "2",
                        "-c", os.path.join(root, "config", "config.yaml")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "########## Inference Metrics ##########" in out
    for name in ("Loss", "R2", "dIoU", "Precision", "Recall", "F1"):
        assert any(line.startswith(name + " .") for line in out.splitlines()), name


def test_streaming_filter_pipeline(net, params):
    """sps_node.callback minus ROS: pose transform -> submap -> infer -> keep score <= eps."""
    from sps_amd.pipeline import StableFilter
    import sps.datasets.util as util
    rng = np.random.default_rng(8)
    map_pts = synthetic.build_map(n_azimuth=400, n_beams=32)
    scan_world = synthetic.lidar_scan(77, x_offset=1.0, n_azimuth=400, n_beams=32)[:, :3].astype(np.float64)
    ang = 0.4
    pose = np.array([[np.cos(ang), -np.sin(ang), 0, 1.0], [np.sin(ang), np.cos(ang), 0, -0.5], [0, 0, 1, 0.0], [0, 0, 0, 1.0]])
    sensor = util.inverse_transform_point_cloud(scan_world, pose)
    f = StableFilter(net, torch.from_numpy(map_pts), voxel_size=VS, epsilon=EPS)
    res = f(sensor, pose)
    # oracle: same stages on the CPU
    world = util.transform_point_cloud(sensor, pose).astype(np.float32)
    sub, n_sv = O.prune(O.to_coords(map_pts[:, :3], VS), O.to_coords(world, VS), VS)
    assert (res.n_scan_voxels, res.n_submap_voxels) == (n_sv, len(sub))
    batch = synthetic.assemble(np.c_[world, np.zeros(len(world))].astype(np.float32), sub)
    ref, _ = O.sps_forward(params, batch[:, :5], VS)
    n = len(world)
    np.testing.assert_allclose(res.scores.cpu().numpy(), ref[:n], rtol=0, atol=1e-4)
    keep = ref[:n] <= np.float32(EPS)
    band = np.abs(ref[:n] - np.float32(EPS)) > 1e-5
    got_keep = (res.scores.cpu().numpy() <= np.float32(EPS))
    np.testing.assert_array_equal(got_keep[band], keep[band])
    assert res.filtered.shape == (int(got_keep.sum()), 3)
    np.testing.assert_allclose(res.filtered.cpu().numpy(), sensor[got_keep].astype(np.float32), rtol=0, atol=0)
    assert res.t_total >= res.t_infer > 0 and res.t_prune > 0
    util._MAP_CACHE.clear()


def test_variant_a_radius_submap_matches_scipy():
    """select_closest_points on the device vs scipy cKDTree.query_ball_tree (blt_dataset.py:258-271):
    per scan point the same hit multiset; the assembled item equals the reference-layout item."""
    from scipy.spatial import cKDTree
    import sps.datasets.blt_dataset as blt
    rng = np.random.default_rng(12)
    pc_map = np.concatenate([rng.uniform(-5, 5, (40000, 3)), rng.uniform(0, 1, (40000, 1))], 1)      # float64
    idx = rng.choice(len(pc_map), 3000, replace=False)
    scan = np.concatenate([pc_map[idx, :3] + rng.normal(0, 0.05, (3000, 3)), rng.uniform(0, 1, (3000, 1))], 1)
    scan[:50, :3] = pc_map[idx[:50], :3]                      # exact coincidences
    scan[50:60, :3] += 100.0                                  # far away: empty hit lists
    r = 0.1
    want = cKDTree(scan[:, :3]).query_ball_tree(cKDTree(pc_map[:, :3]), r)
    sub = blt.DeviceRadiusSubmap(pc_map[:, :3], r)
    got, counts = sub.query(scan[:, :3])
    got, counts = got.cpu().numpy(), counts.cpu().numpy()
    assert counts.tolist() == [len(w) for w in want]
    off = np.concatenate([[0], np.cumsum(counts)])
    for i, w in enumerate(want):
        assert sorted(got[off[i]:off[i + 1]].tolist()) == sorted(w), i
    # assembled item == reference layout (rows of a scan point's list sorted by map index)
    cfg = {"TRAIN": {"AUGMENTATION": False}, "MODEL": {"VOXEL_SIZE": r}}
    ds = blt.BacchusDataset(cfg, [scan], pc_map)
    item = blt.device_item(ds, 0, sub).cpu().numpy()
    assert item.shape == (len(scan) + len(got), 5)
    np.testing.assert_array_equal(item[:len(scan), :3], scan[:, :3].astype(np.float32))
    np.testing.assert_array_equal(item[len(scan):, :3], pc_map[got, :3].astype(np.float32))
    assert (item[:len(scan), 3] == 1).all() and (item[len(scan):, 3] == 0).all() and (item[len(scan):, 4] == 1).all()
    # same voxel set as the reference item -> same scores
    ref_item = ds[0].numpy()
    a = np.unique(O.quantize(np.pad(item[:, :4], ((0, 0), (1, 0))), r), axis=0)
    b = np.unique(O.quantize(np.pad(ref_item[:, :4], ((0, 0), (1, 0))), r), axis=0)
    np.testing.assert_array_equal(a, b)


def _structured_cloud(rng, kind):
    """Small adversarial clouds: [N,6] float32 (b,x,y,z,t,label)."""
    if kind == "dense_cube":                     # every voxel of a 12^3 cube, two time slices: all 81 offsets occur
        g = np.stack(np.meshgrid(*[np.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.1 + 0.05
        xyz = np.concatenate([g, g[: len(g) // 2]], 0) - np.array([0.37, 0.61, 0.2])
        t = np.concatenate([np.ones(len(g)), np.zeros(len(g) // 2)])
    elif kind == "line":                         # a 1-voxel-wide diagonal line: almost no neighbours
        s = np.linspace(-3, 3, 400)
        xyz = np.stack([s, 0.7 * s, 0.2 * s], 1)
        t = (np.arange(400) % 2).astype(float)
    elif kind == "time_slices":                  # t in {-1,0,1,2}: dt = +-1 taps between every pair of slices (4DMOS-like)
        xyz = rng.uniform(-1.5, 1.5, (1500, 3))
        t = rng.integers(-1, 3, 1500).astype(float)
    elif kind == "far_corner":                   # near the limits of the 18-bit voxel key (+-13.1 km at 0.1 m)
        xyz = rng.uniform(-1.0, 1.0, (1200, 3)) + np.array([13090.0, -13090.0, 13000.0])
        t = rng.integers(0, 2, 1200).astype(float)
    elif kind == "duplicates":                   # 40 distinct points repeated 50 times, shuffled
        base = rng.uniform(-0.5, 0.5, (40, 3))
        idx = rng.integers(0, 40, 2000)
        xyz = base[idx]
        t = (idx % 2).astype(float)
    else:                                        # "batches": three batch indices with very different sizes
        xyz = rng.uniform(-2, 2, (2100, 3))
        t = rng.integers(0, 2, 2100).astype(float)
    n = len(xyz)
    b = np.zeros(n)
    if kind == "batches":
        b = np.concatenate([np.zeros(2000), np.ones(90), np.full(10, 2)])
    out = np.zeros((n, 6), np.float32)
    out[:, 0] = b
    out[:, 1:4] = xyz
    out[:, 4] = t
    out[:, 5] = rng.uniform(0, 1, n)
    return out


@pytest.mark.parametrize("kind", ["dense_cube", "line", "time_slices", "far_corner", "duplicates", "batches"])
def test_structured_stress_cases(net, params, kind):
    rng = np.random.default_rng({"dense_cube": 1, "line": 2, "time_slices": 3, "far_corner": 4, "duplicates": 5, "batches": 6}[kind])
    batch = _structured_cloud(rng, kind)
    if kind == "far_corner":
        # float32 coordinates at 13 km have ~1 mm resolution: both sides quantise the SAME float32 values
        check_full(net, params, batch, tol=5e-4)
    else:
        check_full(net, params, batch)
    if kind == "dense_cube":
        # interior voxels of the t = 1 cube see all 27 dt = 0 neighbours
        p = ctx().map_pairs(0)
        assert p[40] == ctx().level_counts()[0] and min(p[27:54]) > 0


def test_fused_forward_metrics_matches_separate_calls(net, params):
    """sps_forward_metrics = sps_forward + sps_metrics_dev: identical scores, identical confusion counts, float sums to
    rounding (the accumulation order of the f64 atomics differs); also with several batch indices and empty input."""
    scenes = [plant_threshold_labels(synthetic.make_scene(scan_seed=11 + i, n_azimuth=300, batch_index=i)["batch"])
              for i in range(3)]
    for nb, arr in ((1, scenes[0]), (3, np.concatenate(scenes, 0))):
        dev = torch.from_numpy(arr).cuda()
        s_ref = net(dev)
        sums_ref = np.asarray(net.step_metrics(dev, s_ref, nb), dtype=np.float64)
        table = torch.full((2, nb, 8), -1.0, dtype=torch.float64, device="cuda")
        s_fused, out = net.forward_metrics(dev, nb, table[1])
        torch.cuda.synchronize()
        assert out.data_ptr() == table[1].data_ptr() and (table[0] == -1).all()
        assert torch.equal(s_fused, s_ref)
        got = out.cpu().numpy()
        for row in got:
            assert_nondegenerate(row)
        np.testing.assert_array_equal(got[:, :5], sums_ref[:, :5])                 # count, TP, FP, FN, TN
        for b in range(nb):
            rows = arr[:, 0] == b
            np.testing.assert_array_equal(got[b, 1:5], oracle_confusion(s_ref.cpu().numpy()[rows], arr[rows]))
        np.testing.assert_allclose(got[:, 5:], sums_ref[:, 5:], rtol=1e-12, atol=1e-9)
        want = O.predict_metrics(s_ref.cpu().numpy(), arr, EPS)
        m = metrics_from_all(got)
        for k in ("loss", "r2", "dIoU", "precision", "recall", "f1"):
            assert abs(m[k] - want[k]) < 1e-9, k
    s0, out0 = net.forward_metrics(torch.empty((0, 6), device="cuda"), 2)
    torch.cuda.synchronize()
    assert s0.shape == (0,) and (out0 == 0).all()
    with pytest.raises(ValueError, match="N, 6"):
        net.forward_metrics(torch.zeros((4, 5), device="cuda"))


def metrics_from_all(sums):
    from sps_amd.models.models import metrics_from_sums
    return metrics_from_sums(np.asarray(sums).sum(axis=0))

@pytest.mark.timeout(600)
def test_ranking_beyond_1024_workgroups(net):
    """Single-pass block ranking (k_rank_points / k_rank_blocks_rows, grid_kernels.inc.h): a workgroup sums the aggregates
    of ALL workgroups with a smaller ticket, 1 024 of them per look-back step.  1.3 M scattered points = 1 270 ranking
    workgroups at level 0 AND (nearly every point its own block) at levels 1..4: voxel sets of every level against numpy on
    the quantised coordinates, the inverse map, first-occurrence row order, run-to-run identical scores."""
    rng = np.random.default_rng(3)
    n = 1_300_000
    xyz = (rng.random((n, 3), dtype=np.float32) * np.float32([400.0, 400.0, 40.0]) - np.float32([200.0, 200.0, 20.0]))
    batch = np.zeros((n, 6), np.float32)
    batch[:, 1:4] = xyz
    batch[::3, 4] = 1.0                                    # two time indices, as a scan + submap batch
    dev, s1 = run(net, batch)
    cx = ctx()
    counts = cx.level_counts()
    q = np.floor(xyz / np.float32(VS)).astype(np.int64)    # f32 division, as models.py:21
    t = batch[:, 4].astype(np.int64)
    for lv in range(5):
        c = q >> lv                                         # floor(c / 2^l): App. A.9
        key = ((t * 4096 + (c[:, 2] + 2048)) * 8192 + (c[:, 1] + 4096)) * 8192 + (c[:, 0] + 4096)
        uniq, first = np.unique(key, return_index=True)
        assert counts[lv] == len(uniq), (lv, counts[lv], len(uniq))
        if lv == 0:
            key0, first0 = key, first
    from sps_amd import _native
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    _native.check(_native.lib.sps_get_inverse(cx.handle, inv.data_ptr()))
    inv = inv.cpu().numpy()
    assert inv.min() == 0 and inv.max() == counts[0] - 1
    rep = np.full(counts[0], -1, np.int64)
    rep[inv] = key0                                          # one key per row ...
    np.testing.assert_array_equal(rep[inv], key0)           # ... and every point of a row has it
    assert len(np.unique(rep)) == counts[0]
    vox = get_voxels(0, counts[0])                           # [b, x, y, z, t] per row
    np.testing.assert_array_equal(vox[inv[:2000], 1:4], q[:2000])
    # rows are block-contiguous with the blocks in first-occurrence order: the first point's voxel lives in block 0
    assert inv[0] < 64
    s1 = s1.clone()
    _, s2 = run(net, batch)
    assert torch.equal(s1, s2) and bool(torch.isfinite(s1).all())
    cx.check_errors(0)
