"""TEST INFRASTRUCTURE ONLY -- CPU restatement of one TRAINING step of the SPS network, for the parity tests of the
HIP training path (sps_train_forward / sps_train_backward).  Imported by tests/ only, never by sps_amd.

Follows the reference:
  * SPSNet.common_step / training_step              src/sps/models/models.py:62-76
      scores = model(coordinates); loss = MSELoss(scores[scan_indices], gt_labels[scan_indices])
  * SPSModel.forward                                src/sps/models/models.py:20-30
  * MinkUNetBase.forward, BasicBlock, downsample    src/sps/models/MinkowskiEngine/minkunet.py:161-219,
                                                    resnet.py:96-126, c_ws/src/mapmos/scripts/minkunet.py:65-82
  * ME.MinkowskiBatchNorm = nn.BatchNorm1d over the active rows (train mode: batch statistics, eps 1e-5,
    momentum 0.1), SURVEY App. A.12
The coordinate sets and kernel maps are the numpy oracle's (oracle/sps_oracle.py, pinned there); the arithmetic is
plain torch on the CPU (index_select / matmul / index_add per kernel offset, F.batch_norm(training=True)), so the
gradients come from torch.autograd -- an implementation independent of the hand-written HIP backward.
parity unpinned for the ME conv conventions (ME is absent, see sps_oracle.py); the autograd part is torch's own.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import sps_oracle as O


def _conv(feats, n_out, kmap, W, transpose=False):
    out = torch.zeros((n_out, W.shape[-1]), dtype=feats.dtype)
    for k, (i, o) in enumerate(kmap):
        if transpose:
            i, o = o, i
        if len(i):
            out = out.index_add(0, torch.from_numpy(np.asarray(o, np.int64)),
                                feats.index_select(0, torch.from_numpy(np.asarray(i, np.int64))) @ W[k])
    return out


def _bn(p, name, x, stats):
    w, b = p[name + ".bn.weight"], p[name + ".bn.bias"]
    stats[name] = (x.mean(0).detach(), x.var(0, unbiased=False).detach(), x.shape[0])
    return F.batch_norm(x, None, None, w, b, True, 0.1, O.BN_EPS)


def _block(p, name, x, cm, ts, stats):
    n = len(x)
    y = torch.relu(_bn(p, name + ".norm1", _conv(x, n, cm.k3(ts), p[name + ".conv1.kernel"]), stats))
    y = _bn(p, name + ".norm2", _conv(y, n, cm.k3(ts), p[name + ".conv2.kernel"]), stats)
    if (name + ".downsample.0.kernel") in p:
        r = _bn(p, name + ".downsample.1", x @ _lin(p[name + ".downsample.0.kernel"], x.shape[1], cm.cv), stats)
    else:
        r = x
    return torch.relu(y + r)


def _lin(W, cin, cv):
    """kernel_size = 1 kernel as a [C_in, C_out] matrix (O.lin_kernel, differentiable: a view of the stored tensor)."""
    if O.cv_get(cv, "lin_layout") == "out_in":
        return W.reshape(-1, cin).t()
    return W.reshape(cin, -1)


def train_step(params: dict, batch: np.ndarray, voxel_size: float, dtype=torch.float64, cv=None):
    """One common_step: returns (loss float, scores [N] numpy, grads dict name -> numpy, batch stats dict
    bn name -> (mean, biased var, rows)).  ``batch`` rows are (b,x,y,z,t,label).  ``cv``: ME-convention options
    (oracle/sps_oracle.py); gradients come back in the layout the parameters were given in."""
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=("running" not in k)) for k, v in params.items()}
    q = O.quantize(batch[:, :5], voxel_size)
    vox, inv = O.unique_first(q)
    cm = O.CoordinateManager(vox, cv)
    mirrored = O.cv_get(cv, "transpose_index") == "mirrored"
    stats = {}
    feats = torch.full((len(vox), 1), 0.5, dtype=dtype)
    out = _conv(feats, len(vox), cm.k5(), p["conv0p1s1.kernel"])
    out_p1 = torch.relu(_bn(p, "bn0", out, stats))
    skips = {1: out_p1}
    cur = out_p1
    ts = 1
    for i, name in enumerate(["conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"]):
        km = cm.kdown(ts)
        cur = torch.relu(_bn(p, f"bn{i + 1}", _conv(cur, len(cm.coords[2 * ts]), km, p[name + ".kernel"]), stats))
        ts *= 2
        cur = _block(p, f"block{i + 1}.0", cur, cm, ts, stats)
        skips[ts] = cur
    for i, name in enumerate(["convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"]):
        fine = ts // 2
        Wup = p[name + ".kernel"]
        cur = _conv(cur, len(cm.coords[fine]), cm.kdown(fine), torch.flip(Wup, dims=(0,)) if mirrored else Wup, transpose=True)
        cur = torch.relu(_bn(p, f"bntr{4 + i}", cur, stats))
        ts = fine
        cur = torch.cat([cur, skips[ts]], dim=1)
        cur = _block(p, f"block{5 + i}.0", cur, cm, ts, stats)
    logits = cur @ _lin(p["final.kernel"], cur.shape[1], cv) + p["final.bias"]
    scores = torch.sigmoid(logits[torch.from_numpy(inv), 0])
    scan = torch.from_numpy(np.flatnonzero(batch[:, 4] == 1))
    gt = torch.tensor(batch[:, 5], dtype=dtype)
    loss = F.mse_loss(scores[scan], gt[scan])               # nn.MSELoss (models.py:52, :68)
    loss.backward()
    grads = {k: v.grad.numpy().copy() for k, v in p.items() if v.requires_grad and v.grad is not None}
    return float(loss.detach()), scores.detach().numpy(), grads, {k: (m.numpy(), v.numpy(), n) for k, (m, v, n) in stats.items()}
