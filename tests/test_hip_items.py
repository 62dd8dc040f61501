"""GPU tests of the offline item path on the device: BacchusDataset.__getitem__
(src/sps/datasets/blt_dataset.py:209-271: scan rows + KD-tree radius submap) + collate_fn (:173-182) as ONE stream-ordered
native call per scan (sps_radius_item), consumed by sps_forward_metrics_n, and the real-data branch of scripts/predict.py
(reference scripts/predict.py:40-83) on a synthetic $DATA tree.  The scipy item path is the checker."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from sps_amd import synthetic
from tests.helpers import CFG, net_from_params, state_dict_from_params, straddle_params, write_data_tree

pytestmark = pytest.mark.gpu

VS = CFG["MODEL"]["VOXEL_SIZE"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(seq="20220629"):
    return {"EXPERIMENT": {"ID": "BLT"},
            "DATA": {"SHUFFLE": False, "NUM_WORKER": 0, "SPLIT": {"TRAIN": [], "VAL": [], "TEST": [seq]}},
            "TRAIN": dict(CFG["TRAIN"], BATCH_SIZE=1), "MODEL": {"VOXEL_SIZE": VS}, "FILTER": dict(CFG["FILTER"])}


def _scene(n_map=6000, n_scan=900, seed=3, dtype=np.float64):
    """A map with duplicates-in-radius (dense patches) and scans sampled around map points."""
    rng = np.random.default_rng(seed)
    pc_map = np.c_[rng.uniform(-3, 3, (n_map, 3)), rng.uniform(0, 1, n_map)]
    pc_map[: n_map // 4, :3] = pc_map[n_map // 4: n_map // 2, :3] + rng.normal(0, 0.02, (n_map // 4, 3))   # close pairs
    scans = []
    for i in range(3):
        pts = pc_map[rng.choice(n_map, n_scan + 37 * i, replace=False), :3] + rng.normal(0, 0.04, (n_scan + 37 * i, 3))
        scans.append(np.c_[pts, rng.uniform(0, 1, len(pts))].astype(dtype))
    return pc_map, scans


def _host_item(cfg, scans, pc_map, idx):
    import sps.datasets.blt_dataset as blt
    ds = blt.BacchusDataset(cfg, scans, pc_map)
    return ds[idx].numpy()


def _sorted_rows(a):
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_radius_item_equals_the_reference_item(dtype):
    """sps_radius_item == BacchusDataset.__getitem__ + the batch column: scan rows in order, submap rows as a multiset
    (duplicates kept; scipy's in-list order is its tree traversal order), chained items append behind *n_rows."""
    from sps_amd.datasets.blt_dataset import DeviceRadiusSubmap
    from sps_amd.models.models import get_context
    cfg = _cfg()
    pc_map, scans = _scene(dtype=dtype)
    st = torch.cuda.current_stream().cuda_stream
    cx = get_context(0, st)
    sub = DeviceRadiusSubmap(pc_map[:, :3], VS, ctx=cx)            # keeps the grid alive
    rows = torch.full((20000, 6), -7.0, dtype=torch.float32, device="cuda")
    nrows = torch.zeros(4, dtype=torch.int32, device="cuda")
    want_all = []
    for j, scan in enumerate(scans[:2]):
        dev = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
        cx.radius_item(dev.data_ptr(), dtype == np.float64, 4, len(scan), float(j), None if j == 0 else nrows.data_ptr(),
                       rows.data_ptr(), 6, rows.shape[0], nrows.data_ptr(), st)
        item = _host_item(cfg, scans, pc_map, j)
        want_all.append(np.c_[np.full(len(item), j, np.float32), item])
    torch.cuda.synchronize()
    cx.check_errors(st)
    total = int(nrows[0])
    assert total == sum(len(w) for w in want_all)
    got = rows[:total].cpu().numpy()
    assert (rows[total:] == -7.0).all()
    o = 0
    for j, want in enumerate(want_all):
        n = len(scans[j])
        g = got[o: o + len(want)]
        np.testing.assert_array_equal(g[:n], want[:n])                                  # scan rows: same order, same f32 values
        assert len(want) > n, "the scene must produce a non-empty submap"
        np.testing.assert_array_equal(_sorted_rows(g[n:]), _sorted_rows(want[n:]))      # submap rows: same multiset
        assert len(np.unique(want[n:], axis=0)) < len(want) - n, "the scene must exercise duplicate hits"
        o += len(want)
    del sub


def test_radius_item_overflow_is_reported():
    from sps_amd._native import ERR_ITEMCAP, SpsError
    from sps_amd.datasets.blt_dataset import DeviceRadiusSubmap
    from sps_amd.models.models import get_context
    pc_map, scans = _scene()
    st = torch.cuda.current_stream().cuda_stream
    cx = get_context(0, st)
    sub = DeviceRadiusSubmap(pc_map[:, :3], VS, ctx=cx)
    n = len(scans[0])
    rows = torch.full((n + 10, 6), -7.0, dtype=torch.float32, device="cuda")
    guard = torch.full((64, 6), -7.0, dtype=torch.float32, device="cuda")
    nrows = torch.zeros(4, dtype=torch.int32, device="cuda")
    dev = torch.from_numpy(scans[0]).cuda()
    cx.radius_item(dev.data_ptr(), True, 4, n, 0.0, None, rows.data_ptr(), 6, rows.shape[0], nrows.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(nrows[0]) == n + 10 and (guard == -7.0).all()
    with pytest.raises(SpsError) as e:
        cx.check_errors(st)
    assert e.value.code == ERR_ITEMCAP and "item buffer" in str(e.value)
    cx.check_errors(st)                                             # the flag is cleared by the report
    del sub


def test_empty_scan_as_the_first_item_of_a_fresh_context_and_sticky_errors():
    """(i) An empty scan as the very first item a context sees: the scratch of the scans (where the item's row base is
    written) is allocated on first use whatever n is -- it used to be a device write through a null pointer.
    (ii) One sticky error is reported per synchronising call and only THAT bit is cleared: an item overflow and an
    out-of-range coordinate on the same context both surface, in two calls."""
    from sps_amd import _native
    from sps_amd._native import ERR_ITEMCAP, ERR_RANGE, SpsError
    from sps_amd.datasets.blt_dataset import DeviceRadiusSubmap
    pc_map, scans = _scene()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        st = stream.cuda_stream
        cx = _native.Context(0)                                    # fresh: no item scratch yet
        sub = DeviceRadiusSubmap(pc_map[:, :3], VS, ctx=cx)
        rows = torch.full((64, 6), -7.0, dtype=torch.float32, device="cuda")
        nrows = torch.full((4,), 99, dtype=torch.int32, device="cuda")
        cx.radius_item(0, True, 4, 0, 0.0, None, rows.data_ptr(), 6, rows.shape[0], nrows.data_ptr(), st)
        stream.synchronize()
        cx.check_errors(st)
        assert int(nrows[0]) == 0 and (rows == -7.0).all()
        # (ii) overflow the item buffer, then an out-of-range forward on the same context
        n = len(scans[0])
        small = torch.empty((n + 10, 6), dtype=torch.float32, device="cuda")
        dev = torch.from_numpy(scans[0]).cuda()
        cx.radius_item(dev.data_ptr(), True, 4, n, 0.0, None, small.data_ptr(), 6, small.shape[0], nrows.data_ptr(), st)
        net = net_from_params(O.random_params(seed=0)).cuda().eval().freeze()
        net.model._sync_weights(cx)
        far = torch.tensor([[0, 2.0e4, 0, 0, 1], [0, 0.1, 0.2, 0.3, 1]], dtype=torch.float32, device="cuda")   # 20 km: outside the key range
        sc = torch.empty(2, dtype=torch.float32, device="cuda")
        cx.forward(far.data_ptr(), 5, 2, VS, sc.data_ptr(), st)
        stream.synchronize()
        assert torch.isnan(sc[0]) and not torch.isnan(sc[1])
        codes = []
        for _ in range(2):
            with pytest.raises(SpsError) as e:
                cx.check_errors(st)
            codes.append(e.value.code)
        assert sorted(codes) == sorted([ERR_RANGE, ERR_ITEMCAP]), codes
        cx.check_errors(st)                                        # both reported, both cleared
        del sub
        cx.close()


@pytest.fixture(scope="module")
def net():
    params = straddle_params(O.random_params(seed=0), synthetic.small_scene(seed=11, n_scan=2500))
    return net_from_params(params).cuda().eval().freeze()


def test_engine_submit_scans_matches_host_items(net):
    """ScanEngine.submit_scans (raw scans -> device items -> forward + metric sums, no host sync) == the same loop fed
    with the reference's host-assembled, collated items: identical counts and confusion sums, loss sums to rounding."""
    import sps.datasets.blt_dataset as blt
    from sps_amd.engine import ScanEngine
    cfg = _cfg()
    pc_map, scans = _scene(n_map=20000, n_scan=3000)
    ds = blt.BacchusDataset(cfg, scans, pc_map)
    groups = [[0, 1], [2]]
    eng = ScanEngine(net, 0, streams=2, table_rows=8)
    want = eng.run_sequence([blt.BacchusModule.collate_fn([ds[i] for i in g]) for g in groups][:1], 2)
    want = np.concatenate([want, eng.run_sequence([blt.BacchusModule.collate_fn([ds[2]])], 1)])
    eng.attach_map(pc_map[:, :3], VS)
    eng.row_factor = 4.0            # the scans sit ON map points: ~2.7 item rows per scan point
    eng.prepare_scans(max(sum(len(scans[i]) for i in g) for g in groups))
    eng.reset_table(8)
    for g in groups:
        eng.submit_scans([scans[i] for i in g])
    got = eng.finish().cpu().numpy()
    assert got.shape == (3, 8) and (got[:, 0] == [len(s) for s in scans]).all()
    np.testing.assert_array_equal(got[:, :5], want[:, :5])          # count, TP, FP, FN, TN
    # the submap rows arrive in another order -> another voxel row order -> scores equal to float32 rounding only
    np.testing.assert_allclose(got[:, 5], want[:, 5], rtol=1e-6)     # sum (s - g)^2
    np.testing.assert_allclose(got[:, 6:], want[:, 6:], rtol=1e-12)   # sum g, sum g^2: labels only
    assert (got[:, 1] + got[:, 2] > 0).all() and (got[:, 3] + got[:, 4] > 0).all()
    # a group that outgrows the item buffers is reported at the sequence's one synchronisation
    from sps_amd._native import ERR_ITEMCAP, SpsError
    eng.row_factor = 1.0
    eng._rows = [None] * len(eng.streams)
    eng.reset_table(8)
    eng.submit_scans([scans[0]])
    with pytest.raises(SpsError) as e:
        eng.finish()
    assert e.value.code == ERR_ITEMCAP


def _run(cmd, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


@pytest.mark.timeout(1200)
def test_predict_cli_real_data_branch_device_items_match_host_items(tmp_path):
    """The reference's command line on its real-data branch (scripts/predict.py:40-83: -w CKPT -seq SEQ, $DATA tree ->
    BacchusModule -> per-scan loop -> six lines), in a child process: the device item path and the scipy item path print
    the same six lines, for batch sizes 1 and 4."""
    import yaml
    seq = "20220629"
    write_data_tree(str(tmp_path), n_scans=6, seq=seq, n_map=30000, n_pts=2500)
    cfg = _cfg(seq)
    cfg_path = tmp_path / "config.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    params = straddle_params(O.random_params(seed=0), synthetic.small_scene(seed=11, n_scan=2500))
    ckpt = tmp_path / "w.ckpt"
    torch.save({"state_dict": {k: torch.as_tensor(v) for k, v in state_dict_from_params(params).items()},
                "hyper_parameters": cfg}, ckpt)
    outs = {}
    for name, extra in (("device-b1", ["-b", "1"]), ("device-b4", ["-b", "4"]), ("host-b4", ["-b", "4", "--host-items"])):
        r = _run([sys.executable, os.path.join("scripts", "predict.py"), "-w", str(ckpt), "-seq", seq, "-c", str(cfg_path),
                  "--timing", "--streams", "3"] + extra, {"DATA": str(tmp_path)})
        assert r.returncode == 0, r.stderr[-3000:]
        lines = {l.split(" ")[0]: l for l in r.stdout.splitlines()}
        assert "timing:" in lines and "6 scans" in lines["timing:"], r.stdout
        outs[name] = [lines[k] for k in ("Loss", "R2", "dIoU", "Precision", "Recall", "F1")]
    assert outs["device-b1"] == outs["device-b4"] == outs["host-b4"], outs
    assert float(outs["host-b4"][2].split()[-1]) > 0, "degenerate dIoU"


def _batches_close(dev_batches, host_batches, augmented):
    assert len(dev_batches) == len(host_batches)
    for d, h in zip(dev_batches, host_batches):
        d, h = d.cpu().numpy(), h.numpy()
        assert d.shape == h.shape
        for b in np.unique(h[:, 0]):
            db, hb = d[d[:, 0] == b], h[h[:, 0] == b]
            ns = int((hb[:, 4] == 1).sum())
            np.testing.assert_array_equal(db[:, [0, 4, 5]][:ns], hb[:, [0, 4, 5]][:ns])
            if not augmented:
                np.testing.assert_array_equal(db[:ns], hb[:ns])
                np.testing.assert_array_equal(_sorted_rows(db[ns:]), _sorted_rows(hb[ns:]))
            else:   # float32 matmul on the GPU vs on the CPU: rounding only; the submap rows arrive in another order
                np.testing.assert_allclose(db[:ns, 1:4], hb[:ns, 1:4], rtol=0, atol=2e-5)
                assert len(db) == len(hb)
                np.testing.assert_allclose(np.sort(db[ns:, 1:4], 0), np.sort(hb[ns:, 1:4], 0), rtol=0, atol=2e-5)


@pytest.mark.parametrize("augment", [False, True])
def test_device_item_loader_reproduces_the_dataloader(augment):
    """DeviceItemLoader == DataLoader(BacchusDataset, batch_size, shuffle, collate_fn) with num_workers = 0 under the same
    seed: same shuffled order (RandomSampler's draws), same items, same augmentation draws (blt_dataset.py:102-118,240-278)."""
    import sps.datasets.blt_dataset as blt
    from torch.utils.data import DataLoader
    cfg = _cfg()
    cfg["TRAIN"] = dict(cfg["TRAIN"], BATCH_SIZE=2, AUGMENTATION=augment)
    cfg["DATA"] = dict(cfg["DATA"], SHUFFLE=True)
    pc_map, scans = _scene(n_map=8000, n_scan=700)
    scans = scans + [s[::-1].copy() for s in scans[:2]]                       # 5 items -> 3 batches, the last one partial
    ds = blt.BacchusDataset(cfg, scans, pc_map, split="train")
    torch.manual_seed(5)
    host = list(DataLoader(ds, batch_size=2, shuffle=True, collate_fn=blt.BacchusModule.collate_fn, num_workers=0))
    after_host = torch.rand(1).item()
    dl = blt.DeviceItemLoader(cfg, scans, pc_map, split="train", shuffle=True, device="cuda:0")
    torch.manual_seed(5)
    dev = list(dl)
    assert torch.rand(1).item() == after_host, "the loader must consume the global generator exactly as the DataLoader does"
    assert len(dl) == 3 and len(dl.dataset) == 5
    _batches_close(dev, host, augment)
    # two ranks: every rank takes every second batch of the same order; the tail is dropped for equal step counts
    torch.manual_seed(5)
    r0 = list(blt.DeviceItemLoader(cfg, scans, pc_map, split="train", shuffle=True, device="cuda:0", shard=(0, 2), even_shards=True))
    torch.manual_seed(5)
    r1 = list(blt.DeviceItemLoader(cfg, scans, pc_map, split="train", shuffle=True, device="cuda:0", shard=(1, 2), even_shards=True))
    assert len(r0) == len(r1) == 1
    _batches_close(r0 + r1, host[:2], augment)


@pytest.mark.timeout(1200)
def test_train_cli_on_a_data_tree_with_device_items(tmp_path):
    """scripts/train.py on a $DATA tree (reference scripts/train.py:21-58): device items by default, --host-items for the
    reference's DataLoader path; both train one epoch, write a checkpoint and report finite losses."""
    import yaml
    write_data_tree(str(tmp_path), n_scans=4, seq="train0", n_map=20000, n_pts=1500)
    import shutil
    for seq in ("val0",):
        shutil.copytree(tmp_path / "sequence" / "train0", tmp_path / "sequence" / seq)
    cfg = _cfg()
    cfg["DATA"]["SPLIT"] = {"TRAIN": ["train0"], "VAL": ["val0"], "TEST": ["val0"]}
    cfg["TRAIN"] = dict(cfg["TRAIN"], BATCH_SIZE=2, AUGMENTATION=True, MAX_EPOCH=1)
    cfg_path = tmp_path / "config.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    for extra in ([], ["--host-items"]):
        out = tmp_path / ("out_host" if extra else "out_dev")
        r = _run([sys.executable, os.path.join("scripts", "train.py"), "-c", str(cfg_path), "--max-epochs", "1", "--out", str(out)] + extra,
                 {"DATA": str(tmp_path)})
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("epoch 000")][0]
        tl, vl = float(line.split("train_loss")[1].split()[0]), float(line.split("val_loss")[1].split()[0])
        assert np.isfinite(tl) and np.isfinite(vl) and 0 < tl < 1 and 0 < vl < 1, line
        assert any(f.endswith(".ckpt") for f in os.listdir(out / "BLT" / "checkpoints"))
