"""CPU-only tests: golden vectors captured from the reference python (tools/capture_goldens.py), the
host-side mirror of the reference API, the C ABI exports, the roofline bookkeeping."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from tests.helpers import CFG, net_from_params, state_dict_from_params

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------ goldens from the reference
def _metric_cases():
    z = np.load(os.path.join(GOLD, "calculate_metrics.npz"))
    return [(z[f"gt{i}"], z[f"pred{i}"], z[f"out{i}"]) for i in range(int(z["n"]))]


def test_calculate_metrics_matches_reference_goldens():
    import sps.datasets.util as util
    from sps_amd.models.models import metrics_from_sums
    for gt, pred, want in _metric_cases():
        with np.errstate(all="ignore"):
            got_oracle = np.asarray(O.calculate_metrics(gt, pred), np.float64)
            got_host = np.asarray(util.calculate_metrics(gt, pred), np.float64)
        np.testing.assert_allclose(got_oracle, want, rtol=0, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(got_host, want, rtol=0, atol=1e-12, equal_nan=True)
        # the device path returns confusion counts; the host turns them into the same five numbers
        tp = np.sum((gt == 1) & (pred == 1)); tn = np.sum((gt == 0) & (pred == 0))
        fp = np.sum((gt == 0) & (pred == 1)); fn = np.sum((gt == 1) & (pred == 0))
        m = metrics_from_sums([len(gt), tp, fp, fn, tn, 0.0, 0.0, 0.0])
        got = np.array([m["precision"], m["recall"], m["f1"], m["accuracy"], m["dIoU"]], np.float64)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12, equal_nan=True)


def test_transform_point_cloud_matches_reference_goldens():
    import sps.datasets.util as util
    z = np.load(os.path.join(GOLD, "transform.npz"))
    np.testing.assert_array_equal(util.transform_point_cloud(z["pts"], z["T"]), z["out_T"])
    np.testing.assert_array_equal(util.transform_point_cloud(z["pts"], z["P"]), z["out_P"])
    np.testing.assert_allclose(util.inverse_transform_point_cloud(z["out_T"], z["T"]), z["inv_T"], rtol=0, atol=1e-12)


def test_bacchus_dataset_and_collate_match_reference_goldens():
    import sps.datasets.blt_dataset as blt
    z = np.load(os.path.join(GOLD, "bacchus_dataset.npz"))
    cfg = {"TRAIN": {"AUGMENTATION": False, "BATCH_SIZE": 2}, "MODEL": {"VOXEL_SIZE": 0.1},
           "DATA": {"NUM_WORKER": 0, "SHUFFLE": False}}
    ds = blt.BacchusDataset(cfg, [z["scan0"], z["scan1"]], z["pc_map"])
    items = [ds[0], ds[1]]
    assert items[0].dtype == torch.float32
    np.testing.assert_array_equal(items[0].numpy(), z["item0"])          # same rows, same order, duplicates kept
    np.testing.assert_array_equal(items[1].numpy(), z["item1"])
    batch = blt.BacchusModule.collate_fn(items)
    np.testing.assert_array_equal(batch.numpy(), z["batch"])
    # the oracle's forward accepts exactly this layout
    assert batch.shape[1] == 6 and set(np.unique(z["batch"][:, 4])) == {0.0, 1.0}


# ------------------------------------------------------------------ host mirror of the reference API
def test_state_dict_keys_and_strict_load():
    p = O.random_params(seed=0)
    net = net_from_params(p)                                  # strict load of reference-named keys
    sd = net.state_dict()
    assert len(sd) == 194
    assert sd["model.MinkUNet.conv0p1s1.kernel"].shape == (125, 1, 8)
    assert sd["model.MinkUNet.block2.0.downsample.0.kernel"].shape == (8, 16)       # 1x1 kernels are 2-D
    assert sd["model.MinkUNet.final.bias"].shape == (1, 1)
    assert sd["model.MinkUNet.convtr4p16s2.kernel"].shape == (8, 64, 64)
    assert "model.MinkUNet.block1.0.downsample.0.kernel" not in sd                  # resnet.py:98
    n_conv = sum(v.numel() for k, v in sd.items() if k.endswith(".kernel"))
    assert n_conv == 1_845_168
    bad = state_dict_from_params(p); bad.pop("model.MinkUNet.bn0.bn.weight")
    from sps_amd.models.models import SPSNet
    with pytest.raises(RuntimeError):
        SPSNet(CFG).load_state_dict(bad)


def test_no_cpu_fallback_and_argument_checks():
    net = net_from_params(O.random_params(seed=0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(4, 6))
    import sps.datasets.util as util
    with pytest.raises(AssertionError, match="cfg is None"):
        util.load_model(None, "x")
    with pytest.raises(AssertionError, match="feature_type need to be either"):
        util.to_coords_features(torch.zeros(3, 3), "foo", 0.1, device="cpu")
    cf = util.to_coords_features(torch.tensor([[0.26, -0.26, 1.0]]), "scan", 0.1, device="cpu")
    assert cf.cloud_coords.tolist() == [[2, -2, 10]] and cf.features.tolist() == [[1.0, 0.0]]   # trunc, one-hot
    assert util.SCAN_TIMESTAMP == 1 and util.MAP_TIMESTAMP == 0
    t = util.add_timestamp(torch.zeros(2, 3), 1, "cpu")
    assert t.shape == (2, 4) and t[:, 3].tolist() == [1.0, 1.0]


def test_metrics_from_sums_matches_oracle_predict_metrics():
    from sps_amd.models.models import metrics_from_sums
    rng = np.random.default_rng(3)
    n = 500
    batch = np.zeros((n, 6), np.float32)
    batch[:, 4] = rng.integers(0, 2, n)
    batch[:, 5] = rng.uniform(0, 1, n).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    want = O.predict_metrics(scores, batch, 0.84)
    scan = batch[:, 4] == 1
    s, g = scores[scan], batch[scan, 5]
    e = np.float32(0.84)
    pred, gt = (s >= e).astype(int), (g >= e).astype(int)
    sums = [scan.sum(), np.sum((gt == 1) & (pred == 1)), np.sum((gt == 0) & (pred == 1)),
            np.sum((gt == 1) & (pred == 0)), np.sum((gt == 0) & (pred == 0)),
            np.sum((s.astype(np.float64) - g) ** 2), np.sum(g.astype(np.float64)), np.sum(g.astype(np.float64) ** 2)]
    got = metrics_from_sums(sums)
    for k in ("loss", "r2", "precision", "recall", "f1", "accuracy", "dIoU"):
        assert got[k] == pytest.approx(want[k], rel=1e-9, abs=1e-12), k


def test_synthetic_scene_is_deterministic_and_sized():
    from sps_amd import synthetic
    a = synthetic.small_scene(seed=3, n_scan=500)
    b = synthetic.small_scene(seed=3, n_scan=500)
    np.testing.assert_array_equal(a, b)
    assert a.dtype == np.float32 and a.shape[1] == 6
    assert (a[:500, 4] == 1).all() and (a[500:, 4] == 0).all() and (a[500:, 5] == 1).all()
    sub = a[500:, 1:4]
    np.testing.assert_array_equal(O.quantize(np.pad(sub, ((0, 0), (1, 1))), 0.1)[:, 1:4] >= -10**6, True)


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/sps_hip.h declares."""
    from sps_amd import _native
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "sps_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sps_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_native._build.LIB)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libsps_hip.so does not export {name}"
    assert declared == set(_native.EXPORTS)
    assert _native.lib.sps_version() >= 100
    assert _native.lib.sps_weights_numel() == 1_848_785
    # error path without a GPU: bad arguments are rejected before any device call
    assert _native.lib.sps_weights_tensor_info(9999, None, 0, None, None) < 0
    assert b"out of range" in _native.lib.sps_last_error()


def test_roofline_bookkeeping_reproduces_survey_numbers():
    """SURVEY.md App. C sizes -> B_alg ~= 281 MB, F_alg ~= 6.17 GFLOP."""
    from sps_amd import roofline
    V = [92808, 43050, 14990, 4735, 1628]
    pairs3 = [1783098, 932350, 298408, 66931, 24460]
    w = roofline.algorithmic_work(135000, V, pairs3, 2410226)
    assert w["flops"] == pytest.approx(6.17e9, rel=0.01)
    assert w["bytes"] == pytest.approx(281e6, rel=0.03)
    assert len(w["per_layer"]) == 33


# ------------------------------------------------------------------ $DATA tree reader (blt_dataset.py:26-100)
from tests.helpers import write_data_tree as _write_data_tree  # noqa: E402


def test_bacchus_module_reads_data_tree(tmp_path, monkeypatch):
    import sps.datasets.blt_dataset as blt
    import sps.datasets.util as util
    pc_map, T_map, scans, poses = _write_data_tree(str(tmp_path))
    monkeypatch.setenv("DATA", str(tmp_path))
    cfg = {"DATA": {"SHUFFLE": False, "NUM_WORKER": 0, "SPLIT": {"TRAIN": [], "VAL": [], "TEST": ["20220629"]}},
           "TRAIN": {"MAP": "base_map.asc.npy", "BATCH_SIZE": 2, "AUGMENTATION": False}, "MODEL": {"VOXEL_SIZE": 0.1}}
    dm = blt.BacchusModule(cfg, test=True)
    assert dm.map.shape == (3000, 4) and len(dm.test_scans) == 3
    # cash_scans: xyz <- T_map . (T_pose . xyz), label column untouched
    for got, scan, pose in zip(dm.test_scans, scans, poses):
        want = util.transform_point_cloud(util.transform_point_cloud(scan[:, :3], pose), T_map)
        np.testing.assert_allclose(got[:, :3], want, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(got[:, 3], scan[:, 3])
    dm.setup()
    batches = list(dm.test_dataloader())
    assert len(batches) == 2 and batches[0].shape[1] == 6
    assert set(batches[0][:, 0].tolist()) == {0.0, 1.0} and set(batches[1][:, 0].tolist()) == {0.0}
    first = batches[0][batches[0][:, 0] == 0]
    assert (first[:120, 4] == 1).all() and (first[120:, 4] == 0).all() and (first[120:, 5] == 1).all()
    # the scans were generated from map points: the radius submap must be non-empty
    assert len(first) > 120
    with pytest.raises(AssertionError, match="should be the same"):
        os.remove(sorted((tmp_path / "sequence" / "20220629" / "poses").iterdir())[0])
        blt.BacchusModule(cfg, test=True)


def test_reference_module_paths_and_map_loader(tmp_path, monkeypatch):
    """The import paths the reference's callers use (src/sps/models/models.py:10, c_ws/src/mos4d/scripts/mos4d.py:9,
    c_ws/src/sps_filter/scripts/sps_node.py:69) resolve to the drop-in package; util.load_point_cloud_map reads
    $DATA/maps/<TRAIN.MAP> as the reference does (.npy and text), float32 [M, 3]."""
    from sps.models.MinkowskiEngine.customminkunet import CustomMinkUNet
    from sps_amd.models.minkunet import CustomMinkUNet as Native
    assert CustomMinkUNet is Native
    m = CustomMinkUNet(in_channels=1, out_channels=3, D=4)
    assert m.state_dict()["final.kernel"].shape == (8, 3)
    import sps.datasets.util as util
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(50, 4))
    os.makedirs(tmp_path / "maps")
    np.save(tmp_path / "maps" / "base_map.asc.npy", pts)
    np.savetxt(tmp_path / "maps" / "base_map.asc", pts)
    monkeypatch.setenv("DATA", str(tmp_path))
    for name in ("base_map.asc.npy", "base_map.asc"):
        got = util.load_point_cloud_map({"TRAIN": {"MAP": name}})
        assert got.dtype == torch.float32 and got.shape == (50, 3)
        np.testing.assert_array_equal(got.numpy(), (pts if name.endswith(".npy") else pts.astype(np.float32))[:, :3].astype(np.float32))
    with pytest.raises(AssertionError, match="cfg is None"):
        util.load_point_cloud_map(None)
    with pytest.raises(RuntimeError, match="Failed to load point cloud map"):
        util.load_point_cloud_map({"TRAIN": {"MAP": "missing.npy"}})


def test_augmentation_matches_reference_goldens():
    """sps.datasets.augmentation + BacchusDataset(split="train", AUGMENTATION) under fixed torch seeds reproduce the clouds
    the reference produced (tools/capture_goldens.py): same draws from torch's generator, same float32 arithmetic."""
    import sps.datasets.augmentation as aug
    import sps.datasets.blt_dataset as blt
    z = np.load(os.path.join(GOLD, "augmentation.npz"))
    pts = torch.from_numpy(z["pts"])
    for seed in (0, 1, 7):
        for name in ("rotate_point_cloud", "rotate_perturbation_point_cloud", "random_flip_point_cloud", "random_scale_point_cloud"):
            torch.manual_seed(seed)
            got = getattr(aug, name)(pts.clone()).numpy()
            np.testing.assert_allclose(got, z[f"{name}_{seed}"], rtol=0, atol=2e-6, err_msg=f"{name} seed {seed}")
    d = np.load(os.path.join(GOLD, "bacchus_dataset.npz"))
    cfg = {"TRAIN": {"AUGMENTATION": True, "BATCH_SIZE": 1}, "MODEL": {"VOXEL_SIZE": 0.1}, "DATA": {"NUM_WORKER": 0, "SHUFFLE": False}}
    for seed in (0, 1, 7):
        torch.manual_seed(seed)
        ds = blt.BacchusDataset(cfg, [d["scan0"], d["scan1"]], d["pc_map"], split="train")
        item = ds[0].numpy()
        want = z[f"item_aug_{seed}"]
        assert item.shape == want.shape
        np.testing.assert_allclose(item[:, :3], want[:, :3], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(item[:, 3:], want[:, 3:])                      # time stamp and label untouched
    # no augmentation outside the training split
    ds = blt.BacchusDataset(cfg, [d["scan0"]], d["pc_map"])
    np.testing.assert_array_equal(ds[0].numpy(), d["item0"])


def test_save_vis_dumps(tmp_path):
    """models.py:113-152: file names, column layout, the pooled-score length assertion for batches of more than one scan."""
    import torch
    from sps_amd.models.models import save_vis
    rng = np.random.default_rng(5)
    scan = np.column_stack([np.zeros(7), rng.normal(size=(7, 3)), np.ones(7), rng.random(7)])
    submap = np.column_stack([np.zeros(5), rng.normal(size=(5, 3)), np.zeros(5), rng.random(5)])
    batch = torch.from_numpy(np.vstack([scan, submap]).astype(np.float32))
    scores = torch.from_numpy(rng.random(12).astype(np.float32))
    paths = save_vis(str(tmp_path / "predictions" / "seq"), batch, 3, scores)
    assert [p.split("predictions")[1] for p in paths] == ["/seq/scans/3_0.0.npy", "/seq/maps/3_0.0.npy"]
    s, m = np.load(paths[0]), np.load(paths[1])
    assert s.shape == (7, 5) and m.shape == (5, 4)
    np.testing.assert_array_equal(s[:, :3], batch[:7, 1:4].numpy())
    np.testing.assert_array_equal(s[:, 3], batch[:7, 5].numpy())
    np.testing.assert_array_equal(s[:, 4], scores[:7].numpy())
    np.testing.assert_array_equal(m, np.column_stack([batch[7:, 1:4].numpy(), batch[7:, 5].numpy()]))
    two = torch.cat([batch, torch.cat([torch.ones(12, 1), batch[:, 1:]], dim=1)])
    with pytest.raises(AssertionError):
        save_vis(str(tmp_path / "p2"), two, 0, torch.cat([scores, scores]))


def test_product_library_has_no_diagnostic_switches():
    """The shipped libsps_hip.so reads no environment variable and contains none of the diagnostic switches: they exist in
    private -DSPS_DIAG builds only (tools/*_sweep.sh load those through $SPS_LIB)."""
    import re
    import subprocess
    from sps_amd import _build
    lib = _build.build()
    blob = open(lib, "rb").read()
    for name in (b"SPS_DIAG_SKIP", b"SPS_NO_MERGE", b"SPS_PX", b"SPS_GRID_SCALE", b"SPS_CONV_MAX_WG", b"SPS_GEOM_L",
                 b"SPS_ABLATE", b"SPS_WAVE_TRACE", b"SPS_TRACE_LAYER", b"SPS_WS_"):
        assert name not in blob, name
    assert not re.search(rb"SPS_[A-Z][A-Z_0-9]{3,}", blob), re.findall(rb"SPS_[A-Z][A-Z_0-9]{3,}", blob)[:5]
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_hostplace_parses_cpulists_and_never_raises():
    from sps_amd import hostplace
    assert hostplace._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert hostplace._parse_cpulist("") == set()
    info = hostplace.bind_to_gpu_numa(0)            # no GPU / no sysfs entry here: a note, not an exception
    assert info["bound"] is False and set(info) >= {"pci", "numa_node", "cpus_before", "cpus_after", "bound"}


def test_scalar_log_writes_lightning_csv_layout(tmp_path):
    """sps_amd/scalars.py: the per-step scalars of the reference's loggers (models.py:74-75,80-81; train.py:38,46-50) in
    <root>/<name>/version_<n>/metrics.csv; tensors are only read at flush()."""
    import csv
    from sps_amd.scalars import ScalarLog
    log = ScalarLog(str(tmp_path), "BLT")
    log.log(0, 0, train_loss=torch.tensor(0.5), train_r2=-1.0, **{"lr-Adam": 7e-5})
    log.log(0, 1, val_loss=torch.tensor(0.25), val_r2=0.1)
    assert not os.path.exists(log.path)
    log.flush()
    log.log(1, 1, train_loss=0.125, train_r2=0.0, **{"lr-Adam": 6.9e-5})
    log.flush()
    rows = list(csv.DictReader(open(log.path)))
    assert list(rows[0]) == ["epoch", "step", "train_loss", "train_r2", "lr-Adam", "val_loss", "val_r2"]
    assert [r["train_loss"] for r in rows] == ["0.5", "", "0.125"] and rows[1]["val_loss"] == "0.25" and rows[2]["epoch"] == "1"
    assert log.dir.endswith("version_0") and ScalarLog(str(tmp_path), "BLT").dir.endswith("version_1")


@pytest.mark.timeout(600)
def test_inference_kernels_have_no_serialised_optional_loads():
    """a per-lane `x = cond ? table[i] : 0` compiles to an exec-masked block per load and a vmcnt(0) after it, so
    N independent optional loads cost N memory round trips.  tools/isa_single_loads.py lists the loads that sit alone between
    two vmcnt(0) in the gfx950 ISA; what is left on the inference path are chains that are dependent by nature (hash walks,
    slot -> rank -> row base, the per-supertile entry of k_conv_px).  Ceilings = the counts of round 4's final build + 2."""
    import re
    import shutil
    import subprocess
    import sys
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_single_loads.py"), "k_maps", "k_link_adj", "k_conv_px", "k_conv0_fused",
                          "k_points_to_blocks"], capture_output=True, text=True, check=True).stdout
    counts = {m.group(1): int(m.group(2)) for m in re.finditer(r"^(\S[^:]*): (\d+)", out, re.M)}
    assert counts, out
    ceilings = {"k_maps": 8, "k_link_adj": 16, "k_conv0_fused": 12, "k_points_to_blocks": 2}
    for name, n in counts.items():
        if name.startswith("k_conv_px"):
            assert n <= 4, (name, n, out)     # the supertile's order / count entry (+ nothing in the epilogues)
        elif name in ceilings:
            assert n <= ceilings[name], (name, n, out)


def test_bench_cpu_quota_and_engine_pipeline_defaults():
    """bench.cpu_quota(): what this job may use (affinity / cgroup quota), never more than the host's cores; the engine's
    pipeline counts are multiples of the runtime's 4 hardware queues."""
    import bench
    from sps_amd import engine
    q = bench.cpu_quota()
    assert 1 <= q <= (os.cpu_count() or 1)
    assert engine.DEFAULT_STREAMS % 4 == 0 and engine.SHORT_RUN_STREAMS % 4 == 0 and engine.SHORT_RUN_STREAMS <= engine.DEFAULT_STREAMS


def test_kernel_durations_tool_averages_launch_positions(tmp_path):
    """tools/kernel_durations.py: the launches between two k_points_to_blocks of a rocprofv3 kernel trace are one scan; per launch
    position the mean duration over the steady-state scans; tagged with the hash of the kernel sources bench.py checks."""
    import csv
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    names = ["(anonymous namespace)::k_points_to_blocks(float const*)", "void (anonymous namespace)::k_conv<2, 2>((anonymous namespace)::ConvArgs)",
             "(anonymous namespace)::k_tail(float const*)"]
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        t = 0
        for scan in range(8):
            for j, nm in enumerate(names):
                dur = 1000 * (j + 1) + (100 if scan % 2 else 0)
                w.writerow([nm, t, t + dur])
                t += dur + 500
        w.writerow([names[0], t, t + 1000])                      # the start of a ninth scan closes the eighth
        w.writerow(["at::native::some_torch_kernel", t + 5, t + 6])       # foreign kernels are ignored
    out = tmp_path / "kd.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_durations.py"), str(tmp_path / "prof"), str(out)], check=True,
                   capture_output=True)
    kd = json.load(open(out))
    import bench
    assert kd["csrc_sha"] == bench.csrc_sha()
    assert [n for n, _ in kd["launches"]] == ["k_points_to_blocks", "k_conv<2, 2>", "k_tail"]
    assert [u for _, u in kd["launches"]] == pytest.approx([1.05, 2.05, 3.05])           # us: mean of the alternating durations
    assert kd["sum_us"] == pytest.approx(6.15)


def test_predict_cli_help_lists_the_reference_options_and_backend():
    """The CLI keeps the reference's options (-w / -seq / -c, predict.py:15-39) and names the distributed backend switch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "predict.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for opt in ("--weights", "-w", "--sequence", "-seq", "--config", "-c", "--backend", "--batch-size"):
        assert opt in r.stdout, opt
