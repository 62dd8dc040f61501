"""GPU tests of the stream-ordered pieces around the forward: the pose transform (a1), the device-count forward, the
epsilon-filter compaction (f2), the pipelined ScanEngine that bench.py and scripts/predict.py run, and the multi-rank
bench control flow.  All through the C ABI (ctypes bindings of include/sps_hip.h)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from sps_amd import synthetic
from tests.helpers import CFG, assert_nondegenerate, net_from_params, plant_threshold_labels, straddle_params

pytestmark = pytest.mark.gpu

VS = CFG["MODEL"]["VOXEL_SIZE"]
EPS = CFG["FILTER"]["THRESHOLD"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def params():
    return straddle_params(O.random_params(seed=0), synthetic.small_scene(seed=11, n_scan=2500))


@pytest.fixture(scope="module")
def net(params):
    return net_from_params(params).cuda().eval().freeze()


def ctx():
    from sps_amd.models.models import get_context
    return get_context(0)


def stream():
    return torch.cuda.current_stream().cuda_stream


def test_transform_points_bit_exact_vs_reference_goldens():
    """util.transform_point_cloud on the device against vectors captured from the reference (tools/capture_goldens.py):
    float64 results bit for bit (same fused multiply-add chain as numpy's dgemm), float32 store, float32 input,
    strided input/output, perspective matrix, identity."""
    z = np.load(os.path.join(GOLD, "transform.npz"))
    pts = z["pts"]
    n = len(pts)
    for T, want in ((z["T"], z["out_T"]), (z["P"], z["out_P"])):
        src = torch.from_numpy(pts).cuda()
        out64 = torch.empty((n, 3), dtype=torch.float64, device="cuda")
        ctx().transform_points(src.data_ptr(), True, 3, n, T, out64.data_ptr(), True, 3, stream())
        np.testing.assert_array_equal(out64.cpu().numpy(), want)
        out32 = torch.zeros((n, 7), dtype=torch.float32, device="cuda")               # strided output rows
        ctx().transform_points(src.data_ptr(), True, 3, n, T, out32.data_ptr() + 8, False, 7, stream())
        np.testing.assert_array_equal(out32[:, 2:5].cpu().numpy(), want.astype(np.float32))
        assert (out32[:, :2] == 0).all() and (out32[:, 5:] == 0).all()
        # float32 input in a wider row (a LiDAR driver's x,y,z,intensity): the reference widens to float64 first
        p32 = np.zeros((n, 4), np.float32)
        p32[:, :3] = pts.astype(np.float32)
        from sps_amd.datasets import util
        ref32 = util.transform_point_cloud(p32[:, :3].astype(np.float64), T)
        ctx().transform_points(torch.from_numpy(p32).cuda().data_ptr(), False, 4, n, T, out64.data_ptr(), True, 3, stream())
        np.testing.assert_array_equal(out64.cpu().numpy(), ref32)
    ctx().transform_points(src.data_ptr(), True, 3, n, None, out64.data_ptr(), True, 3, stream())
    np.testing.assert_array_equal(out64.cpu().numpy(), pts)
    ctx().transform_points(src.data_ptr(), True, 3, 0, z["T"], out64.data_ptr(), True, 3, stream())   # empty: no launch


def test_forward_with_device_side_row_count(net, params):
    """sps_forward_n: the row count comes from device memory (<= the bound the grids were sized for); identical bits."""
    batch = synthetic.small_scene(seed=17, n_scan=2600)
    n = len(batch)
    dev = torch.from_numpy(batch).cuda()
    want = net(dev)
    padded = torch.full((2 * n, 6), 7.5e8, dtype=torch.float32, device="cuda")   # rows past the count are never read
    padded[:n] = dev
    for count in (n, n - 777):
        cnt = torch.tensor([3, 4, count, 5], dtype=torch.int32, device="cuda")
        scores = torch.full((2 * n,), -3.0, dtype=torch.float32, device="cuda")
        net.model._sync_weights(ctx())
        ctx().forward_n(padded.data_ptr(), 6, 2 * n, cnt.data_ptr() + 8, VS, scores.data_ptr(), stream())
        ctx().check_errors(stream())                                             # no range error from the padding rows
        assert (scores[count:] == -3.0).all()
        ref = want if count == n else net(dev[:count])
        assert torch.equal(scores[:count], ref)
        assert ctx().level_counts()[0] == len(np.unique(O.quantize(batch[:count, :5], VS), axis=0))


def test_compact_stable_keeps_order_and_drops_nan(net):
    rng = np.random.default_rng(3)
    n = 70_001                                              # several SCAN_BLOCK chunks + a ragged tail
    e = np.float32(EPS)
    s = rng.uniform(0, 1, n).astype(np.float32)
    s[::11] = e                                             # ties: kept (`<=`, sps_node.py:148)
    s[1::11] = np.nextafter(e, np.float32(1))               # just above: dropped
    s[5::97] = np.nan
    rows = rng.normal(size=(n, 4)).astype(np.float32)
    out = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx().compact_stable(torch.from_numpy(s).cuda().data_ptr(), torch.from_numpy(rows).cuda().data_ptr(), 4, 3, n, EPS,
                         out.data_ptr(), cnt.data_ptr(), stream())
    keep = s <= e
    assert int(cnt.item()) == int(keep.sum()) and 0 < keep.sum() < n
    np.testing.assert_array_equal(out[: int(cnt.item())].cpu().numpy(), rows[keep, :3])
    assert (out[int(cnt.item()):] == -1).all()
    ctx().compact_stable(0, 0, 4, 3, 0, EPS, out.data_ptr(), cnt.data_ptr(), stream())
    assert int(cnt.item()) == 0


def test_streaming_filter_is_stream_ordered_and_matches_oracle(net, params):
    """sps_node.callback minus ROS on several scans issued back to back (no host synchronisation between them), float64
    and float32 input, with and without pose; checked against the oracle's stages."""
    from sps_amd.pipeline import StableFilter
    import sps.datasets.util as util
    map_pts = synthetic.build_map(n_azimuth=400, n_beams=32)
    f = StableFilter(net, torch.from_numpy(map_pts), voxel_size=VS, epsilon=EPS)
    ang = 0.4
    pose = np.array([[np.cos(ang), -np.sin(ang), 0, 1.0], [np.sin(ang), np.cos(ang), 0, -0.5], [0, 0, 1, 0.0], [0, 0, 0, 1.0]])
    cases = []
    for i, (dtype, use_pose) in enumerate(((np.float64, True), (np.float32, True), (np.float32, False))):
        world = synthetic.lidar_scan(77 + i, x_offset=1.0 - i, n_azimuth=400, n_beams=32)[:, :3].astype(np.float64)
        sensor = (util.inverse_transform_point_cloud(world, pose) if use_pose else world).astype(dtype)
        cases.append((sensor, pose if use_pose else None))
    pending = [f.submit(sensor, p) for sensor, p in cases]                  # three scans in flight, nothing synchronised
    any_kept = any_dropped = False
    for (sensor, p), pend in zip(cases, pending):
        res = pend.result()
        world = sensor.astype(np.float64) if p is None else util.transform_point_cloud(sensor.astype(np.float64), p)
        world = world.astype(np.float32)
        sub, n_sv = O.prune(O.to_coords(map_pts[:, :3], VS), O.to_coords(world, VS), VS)
        assert (res.n_scan_voxels, res.n_submap_voxels) == (n_sv, len(sub))
        batch = synthetic.assemble(np.c_[world, np.zeros(len(world))].astype(np.float32), sub)
        ref, _ = O.sps_forward(params, batch[:, :5], VS)
        n = len(world)
        got = res.scores.cpu().numpy()
        assert got.shape == (n,)
        np.testing.assert_allclose(got, ref[:n], rtol=0, atol=1e-4)
        band = np.abs(ref[:n] - np.float32(EPS)) > 1e-5
        got_keep = got <= np.float32(EPS)
        np.testing.assert_array_equal(got_keep[band], (ref[:n] <= np.float32(EPS))[band])
        assert res.filtered.shape == (int(got_keep.sum()), 3)
        np.testing.assert_array_equal(res.filtered.cpu().numpy(), sensor[got_keep].astype(np.float32))   # as received
        any_kept |= bool(got_keep.any())
        any_dropped |= bool((~got_keep).any())
        assert res.t_total > 0 and res.t_prune > 0 and res.t_infer > 0
    assert any_kept and any_dropped                              # the filter removed some points and kept others
    # an unrepresentable coordinate surfaces at result(), not silently as a dropped point
    from sps_amd._native import SpsError
    bad = cases[2][0].copy()
    bad[0, 0] = 3.0e4
    with pytest.raises(SpsError):
        f(bad)
    f(cases[2][0])                                               # the flag was cleared: the next scan is fine


def test_scan_engine_matches_per_scan_predict_step(net, params):
    """The pipelined loop (bench.py / scripts/predict.py) == the reference-shaped per-scan loop: same scores, same
    confusion counts per scan, fed from device tensors, from pinned host tensors and from pageable host tensors."""
    from sps_amd.engine import ScanEngine, per_scan_metrics
    scans = [plant_threshold_labels(synthetic.small_scene(seed=80 + i, n_scan=1800 + 100 * i)) for i in range(9)]
    want = []
    for b in scans:
        want.append(net.predict_step(torch.from_numpy(b).cuda(), 0))
    # (the small random test scenes are not LiDAR-like: full-size arenas; the compact ones have their own test below)
    eng = ScanEngine(net, 0, streams=4, max_rows=max(len(b) for b in scans), table_rows=len(scans), compact=False)
    feeds = {"device": [torch.from_numpy(b).cuda() for b in scans],
             "pinned": [torch.from_numpy(b).pin_memory() for b in scans],
             "pageable": [torch.from_numpy(b) for b in scans]}
    torch.cuda.synchronize()
    for name, feed in feeds.items():
        sums = eng.run_sequence(feed)
        assert sums.shape == (len(scans), 8)
        for row, w in zip(sums, want):
            assert_nondegenerate(row)
            np.testing.assert_array_equal(row[:5], [w["count"], w["tp"], w["fp"], w["fn"], w["tn"]])
        for m, w in zip(per_scan_metrics(sums), want):
            assert m["dIoU"] == pytest.approx(w["dIoU"], abs=1e-12) and m["loss"] == pytest.approx(w["loss"], rel=1e-9), name
    # batch = 4 per step (config 3 layout): per-scan rows unchanged
    groups = [synthetic.collate(scans[i: i + 4]) for i in (0, 4)]
    eng.reset_table(8)
    last = None
    for g in groups:
        last = eng.submit(torch.from_numpy(g).pin_memory(), 4)
    sums4 = eng.finish().cpu().numpy()
    for row, w in zip(sums4, want[:8]):
        np.testing.assert_array_equal(row[:5], [w["count"], w["tp"], w["fp"], w["fn"], w["tn"]])
    assert last.shape == (len(groups[1]),)
    with pytest.raises(ValueError, match="metric table too small"):
        eng.submit(feeds["device"][0], 1, row=eng.table.shape[0])


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


@pytest.mark.timeout(1200)
def test_bench_two_ranks_on_one_gpu_gloo():
    """The N > 1 control flow of bench.py (rank/world from the environment, per-rank engines, the metric all-gather
    inside the timed region, max-over-ranks timing, one JSON line from rank 0), launched exactly as the driver does but
    with the gloo backend so that two ranks can share the one GPU of this box.  Child processes: the parent test
    process never hands its GPU state to them."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29531", "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "12", "--warmup", "3",
              "--azimuth", "500", "--streams", "3", "--no-cpu-baseline", "--no-stages"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(2 * 12 / (d["ms_per_step"] * 12e-3), rel=1e-3)         # whole-job scans/s
    assert d["config"]["sharding"] == "dp2" and d["mean_metrics"]["dIoU"] > 0
    assert d["h2d_inclusive"]["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.timeout(900)
def test_bench_driver_protocol_is_steady_state():
    """`--steps 20 --warmup 5` (what the driver runs): nothing but steady-state work inside the timed region, so the
    short run stays close to a longer one (pipeline fill/drain is all that separates them)."""
    out = {}
    for k, w in ((20, 5), (200, 20)):
        r = _run([sys.executable, "bench.py", "--steps", str(k), "--warmup", str(w), "--no-cpu-baseline", "--no-stages",
                  "--no-h2d"])
        assert r.returncode == 0, r.stderr[-3000:]
        out[k] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out[20]["config"]["streams_per_gpu"] == 4            # five timed steps per pipeline, not 20 one-step pipelines
    assert out[200]["config"]["streams_per_gpu"] == 8 and out[200]["config"]["arena_mb_all_contexts"] < 3072
    assert out[20]["value"] > 0.75 * out[200]["value"], (out[20]["value"], out[200]["value"])
    assert out[20]["mean_metrics"]["dIoU"] > 0


@pytest.mark.timeout(900)
def test_predict_cli_pipelined_matches_bench_loop():
    """scripts/predict.py --synthetic runs the engine loop: six metric lines + a timing line, batch 1 and batch 4 give the
    same per-scan means."""
    outs = []
    for bs in ("1", "4"):
        r = _run([sys.executable, os.path.join("scripts", "predict.py"), "--synthetic", "8", "-c", os.path.join("config", "config.yaml"),
                  "--batch-size", bs, "--timing", "--streams", "3"])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = {l.split(" ")[0]: l for l in r.stdout.splitlines()}
        assert "timing:" in lines, r.stdout
        outs.append([lines[k] for k in ("Loss", "R2", "dIoU", "Precision", "Recall", "F1")])
    assert outs[0] == outs[1]


def _same_sums(a, b):
    """Metric sums of two runs of the same scan: the five counts exactly; the three f64 sums (squared error, labels, labels
    squared) meet through f64 atomics of several workgroups -- their last bit follows the arrival order (tools/soak_engine.py)."""
    a, b = np.asarray(a), np.asarray(b)
    np.testing.assert_array_equal(a[..., :5], b[..., :5])
    np.testing.assert_allclose(a[..., 5:], b[..., 5:], rtol=1e-12, atol=0)


def test_engine_leaves_the_callers_context_alone(net, params):
    """One of the engine's pipelines runs on the caller's current stream (the hardware queues are loaded evenly that way,
    DESIGN 3.2) -- on a context of ITS OWN: the shared per-stream context of that stream, which direct calls of the model use,
    stays dense and keeps its neighbour tables; and the engine's results do not depend on which pipeline took a scan."""
    from sps_amd.engine import ScanEngine
    from sps_amd.models.models import get_context
    scans = [synthetic.small_scene(seed=70 + i, n_scan=1500) for i in range(6)]
    eng = ScanEngine(net, 0, streams=4, max_rows=max(len(b) for b in scans), table_rows=len(scans))
    assert eng.streams[0] is eng.main and eng.ctxs[0] is not get_context(0)
    sums = eng.run_sequence([torch.from_numpy(b) for b in scans])
    dev = torch.from_numpy(scans[0]).cuda()
    s, direct = net.forward_metrics(dev, 1)                              # the caller's stream, the shared context
    torch.cuda.synchronize()
    _same_sums(direct.cpu().numpy()[0], sums[0])                          # (scan 0 ran on the engine's private context)
    c = get_context(0)
    assert c.kernel_map(0, 0).shape[0] == 81                             # still a full context: neighbour table at level 0
    eng2 = ScanEngine(net, 0, streams=2, max_rows=max(len(b) for b in scans), table_rows=len(scans))
    _same_sums(eng2.run_sequence([torch.from_numpy(b) for b in scans]), sums)
    eng1 = ScanEngine(net, 0, streams=1, max_rows=max(len(b) for b in scans), table_rows=len(scans))   # strictly serial: the same
    assert eng1.streams == [eng1.main] and eng1.ctxs[0] is not get_context(0)
    _same_sums(eng1.run_sequence([torch.from_numpy(b) for b in scans]), sums)
    assert get_context(0).kernel_map(0, 0).shape[0] == 81


@pytest.mark.timeout(900)
def test_predict_cli_two_ranks_on_one_gpu_gloo():
    """BASELINE config 5's entry point with W > 1: scripts/predict.py under torch.distributed.run with two ranks sharing
    the one GPU of this box (--backend gloo).  Group g -> rank g mod W, the per-scan index bookkeeping, the padded metric
    all-gather (7 scans: rank 0 owns 4, rank 1 owns 3) and rank 0's mean of per-scan metrics (predict.py:80-83) must print
    the same six lines as the one-rank run -- at batch 1 and at batch 2 (4 groups, the last one short)."""
    base = [os.path.join("scripts", "predict.py"), "--synthetic", "7", "-c", os.path.join("config", "config.yaml"), "--streams", "3"]

    def six(stdout):
        lines = {l.split(" ")[0]: l for l in stdout.splitlines()}
        return [lines[k] for k in ("Loss", "R2", "dIoU", "Precision", "Recall", "F1")]

    for port, bs in ((29551, "1"), (29552, "2")):
        one = _run([sys.executable] + base + ["-b", bs])
        assert one.returncode == 0, one.stderr[-3000:]
        two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port)] + base + ["-b", bs, "--backend", "gloo", "--timing"])
        assert two.returncode == 0, two.stderr[-3000:]
        assert two.stdout.count("########## Inference Metrics ##########") == 1          # rank 0 only
        assert six(two.stdout) == six(one.stdout), (bs, two.stdout, one.stdout)
        assert "timing: 7 scans" in two.stdout and "(2 GPU(s)" in two.stdout, two.stdout


@pytest.mark.timeout(1200)
def test_bench_and_predict_start_their_own_ranks():
    """`python bench.py --gpus N` / `scripts/predict.py --gpus N` WITHOUT torchrun (no WORLD_SIZE in the environment): the
    entry point starts `python -m torch.distributed.run --nproc-per-node N` itself as a child process -- before anything of
    its own touched the GPU -- and relays rank 0's output and the exit code.  Two ranks, then THREE sharing this box's one GPU
    (the box's guard admits six processes on the card, the test process and the launchers included: five ranks were killed
    by it; the 8-rank shard + padded gather runs on CPU in tests/test_parallel_gloo.py): 13 scans on 3 ranks = 5,4,4 rows
    per rank, the one-rank run's six lines."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"

    def run(cmd):
        return subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)

    for n in (2, 3):
        r = run([sys.executable, "bench.py", "--gpus", str(n), "--backend", "gloo", "--steps", "7", "--warmup", "2", "--azimuth", "500",
                 "--streams", "2", "--no-cpu-baseline", "--no-stages"])
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["dist_world_size"] == n and d["steps"] == 7 and d["config"]["sharding"] == f"dp{n}"
        assert d["value"] == pytest.approx(n * 7 / (d["ms_per_step"] * 7e-3), rel=1e-3)
    r = run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "0x"])        # a failing child fails the caller
    assert r.returncode != 0

    base = [sys.executable, os.path.join("scripts", "predict.py"), "--synthetic", "13", "-c", os.path.join("config", "config.yaml"),
            "--streams", "2"]

    def six(stdout):
        lines = {l.split(" ")[0]: l for l in stdout.splitlines()}
        return [lines[k] for k in ("Loss", "R2", "dIoU", "Precision", "Recall", "F1")]

    one = run(base)
    assert one.returncode == 0, one.stderr[-3000:]
    many = run(base + ["--gpus", "3", "--backend", "gloo", "--timing"])
    assert many.returncode == 0, many.stderr[-3000:]
    assert many.stdout.count("########## Inference Metrics ##########") == 1
    assert six(many.stdout) == six(one.stdout), (many.stdout, one.stdout)
    assert "timing: 13 scans" in many.stdout and "(3 GPU(s)" in many.stdout, many.stdout


def test_compact_arena_overflow_aborts_reports_and_recovers(net, params):
    """sps_ctx_set_level_fractions: LiDAR-sized level arrays.  A LiDAR-like cloud runs unchanged in a third of the memory;
    a cloud whose coarse levels do not thin out (every point its own voxel at every stride) makes its forward abort on the
    device: NaN scores, SPS_ERR_NOMEM at the next synchronising call, full-size arenas afterwards; the block hashes are
    left clean (the following forwards are bit-identical to a fresh context's)."""
    from sps_amd import _native
    from sps_amd._native import SpsError
    lidar = synthetic.make_scene(scan_seed=21, n_azimuth=500)["batch"]
    rng = np.random.default_rng(4)
    sparse = np.zeros((len(lidar), 6), np.float32)
    sparse[:, 1:4] = rng.uniform(-400, 400, (len(lidar), 3))         # ~one point per 0.1 m voxel AND per 1.6 m voxel
    sparse[:, 4] = rng.integers(0, 2, len(lidar))
    dense_ctx = ctx()
    net.model._sync_weights(dense_ctx)
    want_lidar = net(torch.from_numpy(lidar).cuda()).cpu().numpy()
    want_sparse = net(torch.from_numpy(sparse).cuda()).cpu().numpy()
    assert np.isfinite(want_sparse).all()
    dense_bytes = None
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        from sps_amd.models.models import get_context
        c = get_context(0, st.cuda_stream)
        c.reserve(len(lidar))
        dense_bytes = c.arena_bytes()
        c.set_level_fractions(c.LIDAR_FRACTIONS)
        c.reserve(len(lidar))
        compact_bytes = c.arena_bytes()
        assert compact_bytes < 0.4 * dense_bytes, (compact_bytes, dense_bytes)
        got = net(torch.from_numpy(lidar).cuda())
        c.check_errors(st.cuda_stream)
        np.testing.assert_array_equal(got.cpu().numpy(), want_lidar)                 # same bits in a third of the memory
        bad = net(torch.from_numpy(sparse).cuda())
        with pytest.raises(SpsError) as ei:
            c.check_errors(st.cuda_stream)
        assert ei.value.code == _native.ERR_NOMEM
        assert torch.isnan(bad).all()                                               # the aborted forward: no stale scores
        again = net(torch.from_numpy(sparse).cuda())                                # full-size arenas now
        c.check_errors(st.cuda_stream)
        np.testing.assert_array_equal(again.cpu().numpy(), want_sparse)
        assert c.arena_bytes() >= dense_bytes
        np.testing.assert_array_equal(net(torch.from_numpy(lidar).cuda()).cpu().numpy(), want_lidar)   # hashes were left clean
        c.check_errors(st.cuda_stream)
    # an aborted forward must not poison the NEXT forward of the same (still compact) context even when the host has
    # not looked at the error flag in between: its tail wiped the block hashes
    st2 = torch.cuda.Stream()
    with torch.cuda.stream(st2):
        c2 = get_context(0, st2.cuda_stream)
        c2.set_level_fractions(c2.LIDAR_FRACTIONS)
        c2.reserve(len(lidar))
        bad2 = net(torch.from_numpy(sparse).cuda())
        good2 = net(torch.from_numpy(lidar).cuda())                                 # no check_errors in between
        bad3 = net(torch.from_numpy(sparse).cuda())
        good3 = net(torch.from_numpy(lidar).cuda())
        with pytest.raises(SpsError) as ei:
            c2.check_errors(st2.cuda_stream)
        assert ei.value.code == _native.ERR_NOMEM
        assert torch.isnan(bad2).all() and torch.isnan(bad3).all()
        np.testing.assert_array_equal(good2.cpu().numpy(), want_lidar)
        np.testing.assert_array_equal(good3.cpu().numpy(), want_lidar)
        c2.set_level_fractions(None)
    # the engine's sequence loop retries by itself
    from sps_amd.engine import ScanEngine
    eng = ScanEngine(net, 0, streams=2, max_rows=len(lidar))
    assert eng.compact
    sums = eng.run_sequence([torch.from_numpy(b).cuda() for b in (lidar, sparse, lidar)])
    assert sums.shape == (3, 8) and np.isfinite(sums).all() and sums[0, 0] == (lidar[:, 4] == 1).sum()
    np.testing.assert_array_equal(sums[0, :5], sums[2, :5])
    np.testing.assert_allclose(sums[0, 5:], sums[2, 5:], rtol=1e-12)     # f64 atomics: order-dependent rounding
    assert not eng.compact                                           # fell back to full-size arenas


@pytest.mark.gpu
def test_inference_only_context_rulebook_replaces_neighbour_table(net):
    """sps_ctx_set_inference_only: at the pair-exact levels the rulebook takes the neighbour table's memory; scores are
    bit-identical, the pair counts (now counted from the rulebook) are the same, sps_get_nbr refuses, and the arena shrinks."""
    from sps_amd import _native
    from sps_amd.models.models import get_context
    batch = torch.from_numpy(synthetic.small_scene(seed=23, n_scan=2400)).cuda()
    st = torch.cuda.Stream()
    cx = get_context(0, st.cuda_stream)
    cx.set_inference_only(False)
    with torch.cuda.stream(st):
        ref = net(batch).clone()
    st.synchronize()
    pairs_full = [cx.map_pairs(l) for l in range(5)]
    bytes_full = cx.arena_bytes()
    cx.set_inference_only(True)
    with torch.cuda.stream(st):
        out = net(batch).clone()
    st.synchronize()
    assert torch.equal(out, ref)
    assert [cx.map_pairs(l) for l in range(5)] == pairs_full
    assert cx.arena_bytes() < bytes_full
    V = cx.level_counts()
    nb = torch.empty((81, V[0]), dtype=torch.int32, device="cuda")
    assert _native.lib.sps_get_nbr(cx.handle, 0, nb.data_ptr()) != 0          # no table at a pair-exact level ...
    nb4 = torch.empty((81, V[4]), dtype=torch.int32, device="cuda")
    assert _native.lib.sps_get_nbr(cx.handle, 4, nb4.data_ptr()) == 0         # ... the coarse levels keep theirs
    cx.set_inference_only(False)


@pytest.mark.timeout(900)
def test_rccl_path_runs_at_world_size_one():
    """No multi-GPU node is available to the tests, so the RCCL code path (init_process_group("nccl", device_id=...), the
    metric-table all_gather, barrier, destroy) is executed through librccl at world size 1, launched the way the driver
    launches the N-GPU bench (torch.distributed.run, one rank per GPU) -- bench.py and scripts/predict.py, in child
    processes that touch the GPU only after the launcher has started them."""
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
    r = _run(base + ["--master-port", "29541", "bench.py", "--gpus", "1", "--force-dist", "--steps", "12", "--warmup", "3",
                     "--azimuth", "500", "--streams", "3", "--no-cpu-baseline", "--no-stages"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["mean_metrics"]["dIoU"] > 0
    r = _run(base + ["--master-port", "29542", os.path.join("scripts", "predict.py"), "--synthetic", "6", "-c",
                     os.path.join("config", "config.yaml"), "--force-dist", "--streams", "3"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "dIoU" in r.stdout and "Loss" in r.stdout


@pytest.mark.timeout(600)
def test_binding_imported_before_torch_shares_torchs_hip_runtime():
    """__graft_entry__.build() imports the binding before anything imports torch; smoke() in the same process must still
    find the GPU (one HIP runtime per process: the binding pulls torch in before it maps libsps_hip.so)."""
    code = ("import sys; from sps_amd import _native; assert 'torch' in sys.modules; "
            "cx = _native.Context(0); import torch; assert torch.cuda.is_available(); "
            "x = torch.zeros(8, device='cuda'); torch.cuda.synchronize(); print('one runtime', cx.arena_bytes() >= 0)")
    r = _run([sys.executable, "-c", code])
    assert r.returncode == 0 and "one runtime True" in r.stdout, r.stderr[-2000:]
