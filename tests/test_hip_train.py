"""GPU parity tests of the training path: sps_train_forward / sps_train_backward behind
SPSNet.training_step against torch.autograd on the CPU restatement (oracle/train_oracle.py) -- same coordinate sets and
kernel maps as the inference oracle, gradients from an independent implementation."""
import numpy as np
import pytest
import torch

from oracle import sps_oracle as O
from oracle import train_oracle as T
from sps_amd import synthetic
from tests.helpers import CFG, net_from_params

pytestmark = pytest.mark.gpu
VS = CFG["MODEL"]["VOXEL_SIZE"]


def rel_err(got, want):
    return float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-12))


def native_step(params, batch):
    net = net_from_params(params).cuda().train()
    dev = torch.from_numpy(batch).cuda()
    out = net.training_step(dev, 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    grads = {k.replace("model.MinkUNet.", ""): p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}
    return net, out, grads


@pytest.mark.timeout(600)
def test_training_step_gradients_match_autograd_oracle():
    batch = synthetic.small_scene(seed=3, n_scan=900)
    params = O.random_params(seed=0)
    loss_ref, scores_ref, grads_ref, stats_ref = T.train_step(params, batch, VS)
    net, out, grads = native_step(params, batch)
    assert float(out["loss"].detach()) == pytest.approx(loss_ref, rel=2e-5)
    net.train()
    with torch.enable_grad():
        s = net(torch.from_numpy(batch).cuda()).detach().cpu().numpy()
    np.testing.assert_allclose(s, scores_ref, rtol=0, atol=2e-5)          # train-mode BatchNorm forward
    assert set(grads) == set(grads_ref)
    worst = {}
    for name, want in grads_ref.items():
        got = grads[name].reshape(want.shape)
        assert np.isfinite(got).all(), name
        worst[name] = rel_err(got, want)
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    assert not bad, f"gradient mismatch (relative to the tensor's max): {bad}"
    # every kind of layer carries a non-trivial gradient
    for k in ("conv0p1s1.kernel", "conv1p1s2.kernel", "block1.0.conv1.kernel", "block4.0.downsample.0.kernel",
              "convtr4p16s2.kernel", "block8.0.conv2.kernel", "block5.0.norm2.bn.weight", "bn0.bn.bias", "final.kernel", "final.bias"):
        assert np.abs(grads_ref[k]).max() > 0, k


@pytest.mark.timeout(600)
def test_running_statistics_follow_batchnorm1d():
    batch = synthetic.small_scene(seed=5, n_scan=700)
    params = O.random_params(seed=1)
    _, _, _, stats_ref = T.train_step(params, batch, VS)
    net, _, _ = native_step(params, batch)
    sd = {k.replace("model.MinkUNet.", ""): v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    for bn, (mean, var, n) in stats_ref.items():
        want_m = 0.9 * params[bn + ".bn.running_mean"] + 0.1 * mean
        want_v = 0.9 * params[bn + ".bn.running_var"] + 0.1 * var * (n / max(n - 1, 1))
        np.testing.assert_allclose(sd[bn + ".bn.running_mean"], want_m, rtol=0, atol=2e-5, err_msg=bn)
        np.testing.assert_allclose(sd[bn + ".bn.running_var"], want_v, rtol=2e-4, atol=2e-5, err_msg=bn)
        assert int(sd[bn + ".bn.num_batches_tracked"]) == 1


@pytest.mark.timeout(600)
def test_training_is_deterministic_and_reduces_the_loss():
    batch = synthetic.small_scene(seed=8, n_scan=1500)
    params = O.random_params(seed=2)
    _, o1, g1 = native_step(params, batch)
    _, o2, g2 = native_step(params, batch)
    assert float(o1["loss"].detach()) == float(o2["loss"].detach())
    for k in g1:
        np.testing.assert_array_equal(g1[k], g2[k], err_msg=k)            # fixed-order reductions: same bits every run
    cfg = dict(CFG)
    cfg["TRAIN"] = dict(CFG["TRAIN"], LR=2e-3)
    net = net_from_params(params, cfg).cuda().train()
    (opt,), (sched,) = net.configure_optimizers()
    dev = torch.from_numpy(batch).cuda()
    losses = []
    for step in range(12):
        opt.zero_grad()
        out = net.training_step(dev, step)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    assert losses[-1] < 0.7 * losses[0], losses
    # eval mode afterwards uses the updated parameters and running statistics through the inference path
    net.eval()
    with torch.no_grad():
        s = net(dev)
    assert torch.isfinite(s).all() and s.shape == (len(batch),)
    v = net.validation_step(dev, 0)
    assert set(v) == {"val_loss", "val_r2"}


@pytest.mark.timeout(900)
def test_train_cli_writes_a_checkpoint_predict_cli_loads(tmp_path):
    """scripts/train.py --synthetic: two epochs, loss printed per epoch, Lightning-layout checkpoints; scripts/predict.py -w
    loads the result (predict.py:56-58)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "train.py"), "-c", os.path.join(root, "config", "config.yaml"),
                        "--synthetic", "6", "--max-epochs", "2", "--out", str(tmp_path)], capture_output=True, text=True,
                       timeout=800, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 2 and "val_loss" in lines[0]
    ck = os.path.join(str(tmp_path), "BLT", "checkpoints", "last.ckpt")
    sd = torch.load(ck, map_location="cpu", weights_only=False)["state_dict"]
    assert "model.MinkUNet.block5.0.conv1.kernel" in sd and int(sd["model.MinkUNet.bn0.bn.num_batches_tracked"]) == 12
    # the scalars the reference logs per step (models.py:74-75,80-81, LearningRateMonitor) in <out>/<ID>/version_0/metrics.csv
    import csv
    rows = list(csv.DictReader(open(os.path.join(str(tmp_path), "BLT", "version_0", "metrics.csv"))))
    assert {"epoch", "step", "train_loss", "train_r2", "lr-Adam", "val_loss", "val_r2"} <= set(rows[0])
    tr = [r for r in rows if r["train_loss"]]
    assert len(tr) == 12 and [int(r["step"]) for r in tr] == list(range(12)) and {r["epoch"] for r in tr} == {"0", "1"}
    assert all(np.isfinite(float(r["train_loss"])) and float(r["lr-Adam"]) > 0 for r in tr)
    assert float(tr[-1]["lr-Adam"]) < float(tr[0]["lr-Adam"])             # StepLR: one decay between the two epochs
    assert any(r["val_loss"] for r in rows)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "predict.py"), "-w", ck, "--synthetic", "2",
                        "-c", os.path.join(root, "config", "config.yaml")], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert any(l.startswith("dIoU .") for l in r.stdout.splitlines())


@pytest.mark.timeout(900)
def test_data_parallel_training_two_ranks_one_gpu(tmp_path):
    """scripts/train.py under torchrun with two ranks (gloo, both on this box's one GPU): batches sharded i mod 2, the flat
    gradient averaged by one all-reduce per step, rank 0 writes the checkpoints; with identical data on both ranks the
    averaged gradient equals the single-process gradient, so the weights after one epoch match a single-process run
    that sees each batch once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SPS_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(root, "scripts", "train.py"), "-c", os.path.join(root, "config", "config.yaml"),
                        "--synthetic", "8", "--max-epochs", "1", "--out", str(tmp_path)], capture_output=True, text=True,
                       timeout=800, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 1, r.stdout[-2000:]                                   # rank 0 only
    sd = torch.load(os.path.join(str(tmp_path), "BLT", "checkpoints", "last.ckpt"), map_location="cpu", weights_only=False)["state_dict"]
    assert int(sd["model.MinkUNet.bn0.bn.num_batches_tracked"]) == 4           # 8 batches over 2 ranks
    assert all(torch.isfinite(v).all() for v in sd.values() if v.dtype.is_floating_point)


def test_training_after_the_context_was_compact():
    """A context that an engine switched to LiDAR-sized arenas goes back to full-size ones for training (the backward
    indexes every level on the host's assumption that nothing was aborted), and the training arena follows the
    re-allocation: gradients equal those of a fresh context."""
    from sps_amd.models.models import get_context
    batch = synthetic.small_scene(seed=3, n_scan=900)
    params = O.random_params(seed=0)
    _, out0, g0 = native_step(params, batch)                 # training arena exists now, sized for this cloud
    c = get_context(0)
    c.set_level_fractions(c.LIDAR_FRACTIONS)
    net = net_from_params(params).cuda().eval().freeze()
    s = net(torch.from_numpy(batch).cuda())                  # inference on the compact arena (this cloud overflows it: fine)
    torch.cuda.synchronize()
    try:
        c.check_errors(torch.cuda.current_stream().cuda_stream)
    except Exception:
        pass
    c.set_level_fractions(c.LIDAR_FRACTIONS)                 # compact again, same capacity: arena re-allocated
    _, out1, g1 = native_step(params, batch)
    assert float(out1["loss"].detach()) == float(out0["loss"].detach())
    for k in g0:
        np.testing.assert_array_equal(g0[k], g1[k], err_msg=k)
    c.set_level_fractions(None)


@pytest.mark.timeout(600)
def test_training_follows_replaced_and_moved_tensors():
    """The cached training plan (flat views of the parameters) must notice load_state_dict(assign=True) and .to():
    after either, a step computes exactly what a freshly built model with the same weights computes."""
    batch = synthetic.small_scene(seed=9, n_scan=600)
    dev = torch.from_numpy(batch).cuda()
    net = net_from_params(O.random_params(seed=2)).cuda().train()
    net.training_step(dev, 0)["loss"].backward()                     # builds the plan
    other = net_from_params(O.random_params(seed=4)).cuda().train()
    sd = {k: v.detach().clone() for k, v in other.state_dict().items()}
    net.load_state_dict(sd, assign=True)
    net.zero_grad(set_to_none=True)
    out = net.training_step(dev, 0)
    out["loss"].backward()
    want = other.training_step(dev, 0)
    want["loss"].backward()
    torch.cuda.synchronize()
    assert float(out["loss"].detach()) == float(want["loss"].detach())
    for (k, p), (_, q) in zip(net.named_parameters(), other.named_parameters()):
        assert torch.equal(p.grad, q.grad), k
    # a round trip through the host re-creates every tensor
    net = net.cpu().cuda()
    net.zero_grad(set_to_none=True)
    again = net.training_step(dev, 0)
    torch.cuda.synchronize()
    # the first step updated the BatchNorm running statistics only: the train-mode forward does not read them
    assert float(again["loss"].detach()) == float(out["loss"].detach())


@pytest.mark.timeout(600)
def test_training_step_with_the_reference_batch_of_two_scans():
    """config.yaml trains with BATCH_SIZE 2: two collated scans (batch indices 0 and 1, overlapping coordinates) in one
    step -- loss, train-mode scores and every gradient against the autograd oracle."""
    a = synthetic.small_scene(seed=21, n_scan=500)
    b = synthetic.small_scene(seed=22, n_scan=650)
    batch = synthetic.collate([a, b])
    assert set(np.unique(batch[:, 0])) == {0.0, 1.0}
    params = O.random_params(seed=5)
    loss_ref, scores_ref, grads_ref, _ = T.train_step(params, batch, VS)
    net, out, grads = native_step(params, batch)
    assert float(out["loss"].detach()) == pytest.approx(loss_ref, rel=2e-5)
    bad = {}
    for name, want in grads_ref.items():
        got = grads[name].reshape(want.shape)
        assert np.isfinite(got).all(), name
        e = rel_err(got, want)
        if e > 2e-3:
            bad[name] = e
    assert not bad, f"gradient mismatch (relative to the tensor's max): {bad}"
    # the two scans do not see each other (they only share the BatchNorm statistics): a batch made of the same scan twice
    # gives both copies the same scores, up to the rounding of differently grouped MFMA sums (the copies' rows fall into
    # different 16-row tiles, whose lists of present offsets group the products differently)
    twice = synthetic.collate([a, a])
    net.zero_grad(set_to_none=True)
    with torch.enable_grad():
        s = net(torch.from_numpy(twice).cuda()).detach().cpu().numpy()
    np.testing.assert_allclose(s[: len(a)], s[len(a):], rtol=0, atol=2e-6)


@pytest.mark.timeout(1200)
def test_training_gradients_at_scale_batch_of_two():
    """The reference's training batch (config.yaml: BATCH_SIZE 2; models.py:62-70) at tens of thousands of rows: the
    multi-chunk ordered reductions of k_wgrad, the 256-partial BatchNorm finish and conv0's 4096-wave weight gradient are
    DIFFED against the float64 autograd oracle, not just timed (the small-scene tests never leave their first chunk)."""
    b1 = synthetic.small_scene(seed=3, n_scan=24000, extent=22.0)
    b2 = synthetic.small_scene(seed=4, n_scan=24000, extent=22.0)
    b2[:, 0] = 1
    batch = np.concatenate([b1, b2])
    assert len(batch) >= 60000
    params = O.random_params(seed=0)
    loss_ref, scores_ref, grads_ref, stats_ref = T.train_step(params, batch, VS)
    net, out, grads = native_step(params, batch)
    from sps_amd.models.models import get_context
    V = get_context(0).level_counts()
    assert V[0] >= 40000 and V[4] >= 64, V
    assert float(out["loss"].detach()) == pytest.approx(loss_ref, rel=5e-5)
    worst = {}
    for name, want in grads_ref.items():
        got = grads[name].reshape(want.shape)
        assert np.isfinite(got).all(), name
        worst[name] = rel_err(got, want)
    bad = {k: v for k, v in worst.items() if v > 3e-3}
    assert not bad, f"gradient mismatch (relative to the tensor's max): {bad}"
    sd = {k.replace("model.MinkUNet.", ""): v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    for bn, (mean, var, n) in stats_ref.items():
        np.testing.assert_allclose(sd[bn + ".bn.running_mean"], 0.9 * params[bn + ".bn.running_mean"] + 0.1 * mean,
                                   rtol=0, atol=5e-5, err_msg=bn)


def test_backward_refuses_overwritten_activations():
    """The activations a backward reads live in the native context: a later forward on the same context (a second training
    forward before one backward, an evaluation forward issued mid-step) must make the stale node's backward FAIL, not
    return the gradients of the wrong forward (sps_train_backward_at)."""
    from sps_amd._native import SpsError
    b1 = synthetic.small_scene(seed=3, n_scan=900)
    b2 = synthetic.small_scene(seed=9, n_scan=700)
    net = net_from_params(O.random_params(seed=0)).cuda().train()
    d1, d2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
    l1 = net.training_step(d1, 0)["loss"]
    l2 = net.training_step(d2, 1)["loss"]                       # overwrites the activations of step 1
    l2.backward()                                               # the live forward: fine
    with pytest.raises((SpsError, RuntimeError), match="overwritten"):
        l1.backward()
    net.zero_grad()
    l3 = net.training_step(d1, 2)["loss"]
    net.eval()
    with torch.no_grad():
        net(d2)                                                 # an evaluation forward on the same context, mid-step
    net.train()
    with pytest.raises((SpsError, RuntimeError), match="overwritten"):
        l3.backward()
    # and the normal sequence still works afterwards
    net.zero_grad()
    net.training_step(d1, 3)["loss"].backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_train_mode_batchnorm_refuses_a_single_row():
    """nn.BatchNorm1d (ME.MinkowskiBatchNorm in the reference, resnet.py:100-107) raises on one value per channel in
    training mode; a cloud that collapses to a single voxel at a coarse level is reported, not silently normalised."""
    from sps_amd._native import ERR_INVALID, SpsError
    from sps_amd.models.models import get_context
    rng = np.random.default_rng(0)
    xyz = rng.uniform(0.05, 1.5, (300, 3)).astype(np.float32)   # one voxel at tensor stride 16
    batch = np.c_[np.zeros(300), xyz, np.ones(300), rng.uniform(0, 1, 300)].astype(np.float32)
    net = net_from_params(O.random_params(seed=0)).cuda().train()
    net.training_step(torch.from_numpy(batch).cuda(), 0)
    cx = get_context(0)
    assert cx.level_counts()[4] == 1
    with pytest.raises(SpsError) as e:
        cx.check_errors(torch.cuda.current_stream().cuda_stream)
    assert e.value.code == ERR_INVALID and "more than 1 value" in str(e.value)


def test_scan_mse_equals_the_torch_formulation_of_common_step():
    """sps_scan_mse / sps_scan_mse_backward (the loss of common_step in two + one launches) against nn.MSELoss over the
    rows with t == 1 and torchmetrics' R2Score formula in f64 torch ops -- value, R2 and the gradient wrt the scores;
    a strided batch view and an all-submap batch (no selected row: NaN like the mean of an empty tensor)."""
    from sps_amd.models.models import SPSNet, _ScanMSE
    g = torch.Generator().manual_seed(5)
    n = 70_001
    wide = torch.rand((n, 8), generator=g)
    wide[:, 4] = (torch.rand(n, generator=g) < 0.45).float()
    wide[:, 5] = torch.rand(n, generator=g)
    for batch in (wide[:, :6].contiguous().cuda(), wide.cuda()[:, :7]):
        scores = torch.rand(n, generator=g).cuda().requires_grad_(True)
        loss, r2 = _ScanMSE.apply(scores, batch)
        (3.0 * loss).backward()
        s64 = scores.detach().double().requires_grad_(True)
        sel = batch[:, 4] == 1
        y = batch[sel, 5].double()
        want = torch.mean((s64[sel] - y) ** 2)
        (3.0 * want).backward()
        want_r2 = 1.0 - torch.sum((y - s64[sel].detach()) ** 2) / torch.sum((y - y.mean()) ** 2)
        assert float(loss) == pytest.approx(float(want), rel=1e-6)
        assert float(r2) == pytest.approx(float(want_r2), rel=1e-6, abs=1e-6)
        np.testing.assert_allclose(scores.grad.cpu().numpy(), s64.grad.float().cpu().numpy(), rtol=1e-5, atol=1e-12)
        assert float(scores.grad[~sel].abs().max()) == 0.0
    none = wide[:, :6].clone()
    none[:, 4] = 0
    loss, _ = _ScanMSE.apply(torch.rand(n).cuda(), none.cuda())
    assert torch.isnan(loss)
    assert SPSNet.common_step.__code__.co_names.count("_ScanMSE") == 1


def test_common_step_dispatch_native_loss_equals_the_torch_fallback():
    """common_step itself (models.py:62-70) on both of its paths: a float32 row-major CUDA batch goes through the library's
    loss (sps_scan_mse), a batch with a column stride (a transposed view) through the torch formulation -- same loss, same
    R2, same gradients of the network's parameters; an EMPTY validation batch takes the fallback and yields NaN instead of failing."""
    params = O.random_params(seed=0)
    batch = synthetic.small_scene(seed=13, n_scan=800)
    taken = []
    from sps_amd.models import models as M
    orig = M._ScanMSE.apply

    def spy(*a):
        taken.append(True)
        return orig(*a)

    res = []
    for strided in (False, True):
        net = net_from_params(params).cuda().train()
        dev = torch.from_numpy(batch).cuda()
        if strided:
            dev = dev.t().contiguous().t()                      # same values, stride(1) != 1: the torch fallback
            assert dev.stride(1) != 1
        taken.clear()
        M._ScanMSE.apply = spy
        try:
            loss, r2 = net.common_step(dev)
        finally:
            M._ScanMSE.apply = orig
        assert bool(taken) == (not strided)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), float(r2), {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}))
    (l0, r0, g0), (l1, r1, g1) = res
    assert l0 == pytest.approx(l1, rel=1e-5) and r0 == pytest.approx(r1, rel=1e-4, abs=1e-5)
    for k in g0:
        assert rel_err(g0[k], g1[k]) < 1e-4, k
    net = net_from_params(params).cuda().eval()                 # (validation_step; a TRAINING step refuses an empty batch)
    loss, _ = net.common_step(torch.zeros((0, 6), dtype=torch.float32, device="cuda"))
    assert torch.isnan(loss)
