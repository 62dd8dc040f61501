"""oracle/sps_oracle.py -- CPU restatement (numpy, float32) of the SPS per-scan hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sps_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.

What it restates (reference = /root/reference, ibrahimhroob/SPS):
  * ``SPSModel.forward``                src/sps/models/models.py:20-30
  * ``MinkUNetBase.forward`` wiring     src/sps/models/MinkowskiEngine/minkunet.py:161-219
  * ``ResNetBase._make_layer``          src/sps/models/MinkowskiEngine/resnet.py:96-126
  * ``BasicBlock.forward``              c_ws/src/mapmos/scripts/minkunet.py:65-82
  * ``util.to_coords_features/prune``   src/sps/datasets/util.py:67-114
  * ``util.calculate_metrics``          src/sps/datasets/util.py:285-299
  * ``SPSNet.predict_step`` metrics     src/sps/models/models.py:84-105

PARITY STATUS.  The sparse-tensor arithmetic of the reference lives in the third-party
library NVIDIA/MinkowskiEngine (un-pinned ``git clone`` of master in the reference
Dockerfile:38-40; era-matching release 0.5.4), which is absent from /root/reference and
not installable here.  Its conventions are restated from the library's published
behaviour.  For that part: **parity unpinned** -- the restatement
is instead validated against an independent dense ``torch.nn.functional.conv3d``
formulation (tests/test_oracle_dense_equiv.py) and hand-computed known-answer tests.
The metric / dataset-assembly functions ARE pinned by golden vectors captured from the
stub-imported reference python (tests/golden/, tools/capture_goldens.py).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, used by ME.MinkowskiBatchNorm

# CustomMinkUNet widths: customminkunet.py:11-12
PLANES = (8, 16, 32, 64, 64, 32, 16, 8)
INIT_DIM = 8


# --------------------------------------------------------------------------------------
# coordinates
# --------------------------------------------------------------------------------------
def quantize(coords: np.ndarray, voxel_size: float) -> np.ndarray:
    """models.py:16,21 then ME TensorField.sparse() floor (App. A.1, A.2).

    ``torch.Tensor([1.0, vs, vs, vs, 1.0])`` is float32, the division is float32 true
    division; ME then floors every column (b and t included) to int32.
    """
    q = np.array([1.0, voxel_size, voxel_size, voxel_size, 1.0], dtype=F32)
    c = np.asarray(coords, dtype=F32) / q
    return np.floor(c).astype(np.int32)


def unique_first(rows: np.ndarray):
    """Unique integer rows in FIRST-OCCURRENCE order + inverse map (App. A.3)."""
    rows = np.ascontiguousarray(rows)
    if len(rows) == 0:
        return rows.reshape(0, rows.shape[1]), np.zeros(0, np.int64)
    uniq, first, inv = np.unique(rows, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    order = np.argsort(first, kind="stable")          # sorted-unique id -> first-occurrence rank
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return uniq[order], rank[inv].astype(np.int64)


def stride_coords(coords: np.ndarray, ts: int) -> np.ndarray:
    """Parent coordinates for a stride-2 conv at input tensor stride ``ts`` (App. A.9):
    floor(c / 2ts) * 2ts on x,y,z; b and t untouched."""
    out = coords.copy()
    s2 = 2 * ts
    out[:, 1:4] = np.floor_divide(coords[:, 1:4], s2) * s2
    return out


# The ME conventions SURVEY App. A marks as unverifiable, as OPTIONS of the restatement (same names and values as
# sps_amd/conventions.py, which this module deliberately does not import: the oracle realises an option in the
# GEOMETRY it enumerates, the product as a permutation of the weight blob -- tests hold the two against each other).
# ``cv`` below is any object with these attributes (or None = the defaults of App. A).
CV_DEFAULTS = dict(offset_order="x_fastest", odd_kernel_sign="plus", even_kernel_order="ascending",
                   transpose_index="same", lin_layout="in_out")


def cv_get(cv, name: str) -> str:
    if cv is None:
        return CV_DEFAULTS[name]
    v = cv[name] if isinstance(cv, dict) else getattr(cv, name)
    assert isinstance(v, str), (name, v)
    return v


def kernel_offsets(ksize, ts_xyz: int, cv=None) -> np.ndarray:
    """[K,4] offsets (dx,dy,dz,dt) = in - out that weight slice k is applied to (App. A.6-A.8).

    Defaults: x fastest / t slowest; odd k: centred {-(k-1)/2..(k-1)/2}; even k: {0..k-1}; spatial offsets scaled by
    the input tensor stride, temporal tensor stride is always 1 in this network.  Options (``cv``): the enumeration
    runs t fastest / x slowest; odd axes are walked from + to - (in = out - o_k); even axes from k-1 down to 0.
    """
    kx, ky, kz, kt = ksize
    strides = (ts_xyz, ts_xyz, ts_xyz, 1)
    minus = cv_get(cv, "odd_kernel_sign") == "minus"
    desc = cv_get(cv, "even_kernel_order") == "descending"

    def axis(k, s):
        if k % 2 == 1:
            a = [(i - k // 2) * s for i in range(k)]
            return a[::-1] if minus else a
        a = [i * s for i in range(k)]
        return a[::-1] if desc else a

    ax = [axis(k, s) for k, s in zip((kx, ky, kz, kt), strides)]
    offs = []
    if cv_get(cv, "offset_order") == "x_fastest":
        for it in ax[3]:
            for iz in ax[2]:
                for iy in ax[1]:
                    for ix in ax[0]:
                        offs.append((ix, iy, iz, it))
    else:
        for ix in ax[0]:
            for iy in ax[1]:
                for iz in ax[2]:
                    for it in ax[3]:
                        offs.append((ix, iy, iz, it))
    return np.asarray(offs, dtype=np.int64)


def lin_kernel(W: np.ndarray, cin: int, cout: int, cv=None) -> np.ndarray:
    """The [C_in, C_out] matrix of a ``kernel_size = 1`` convolution (App. A.11) from the stored tensor: 2-D
    [C_in, C_out] (or 3-D [1, C_in, C_out]) by default, the memory read as [C_out, C_in] under lin_layout = out_in."""
    W = np.asarray(W)
    if cv_get(cv, "lin_layout") == "out_in":
        return np.ascontiguousarray(W.reshape(cout, cin).T)
    return W.reshape(cin, cout)


class _CoordIndex:
    """Packed-key index over an integer coordinate set [V,5] (b,x,y,z,t) for lookups."""

    def __init__(self, coords: np.ndarray, pad: int):
        c = coords.astype(np.int64)
        self.lo = c.min(axis=0) - pad if len(c) else np.zeros(5, np.int64)
        hi = c.max(axis=0) + pad if len(c) else np.zeros(5, np.int64)
        self.ext = hi - self.lo + 1
        keys = self._pack(c)
        self.order = np.argsort(keys, kind="stable")
        self.keys = keys[self.order]

    def _pack(self, c):
        k = np.zeros(len(c), np.int64)
        for a in range(5):
            k = k * self.ext[a] + (c[:, a] - self.lo[a])
        return k

    def lookup(self, q: np.ndarray) -> np.ndarray:
        """row index of each query coordinate, -1 when absent."""
        q = q.astype(np.int64)
        inside = np.all((q >= self.lo) & (q < self.lo + self.ext), axis=1)
        res = np.full(len(q), -1, np.int64)
        if len(self.keys) == 0 or not inside.any():
            return res
        k = self._pack(q[inside])
        pos = np.searchsorted(self.keys, k)
        pos = np.minimum(pos, len(self.keys) - 1)
        hit = self.keys[pos] == k
        tmp = np.where(hit, self.order[pos], -1)
        res[inside] = tmp
        return res


def kernel_map(in_coords: np.ndarray, out_coords: np.ndarray, offsets: np.ndarray):
    """ME kernel map (App. A.8): for each offset k, the pairs (in_row, out_row) with
    in_coord == out_coord + offset_k.  Returns a list of K (in_idx, out_idx) arrays."""
    pad = int(np.abs(offsets).max()) + 1 if len(offsets) else 1
    index = _CoordIndex(in_coords, pad)
    maps = []
    oc = out_coords.astype(np.int64)
    for off in offsets:
        q = oc.copy()
        q[:, 1:5] += off
        hit = index.lookup(q)
        o = np.nonzero(hit >= 0)[0]
        maps.append((hit[o], o))
    return maps


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
def sparse_conv(feats: np.ndarray, n_out: int, kmap, W: np.ndarray, transpose=False) -> np.ndarray:
    """Generalised sparse convolution, offsets accumulated in ascending k (App. A.8).
    ``W`` is [K, C_in, C_out].  ``transpose`` swaps the roles of the map's in/out (A.10)."""
    out = np.zeros((n_out, W.shape[2]), dtype=F32)
    for k, (i, o) in enumerate(kmap):
        if transpose:
            i, o = o, i
        if len(i):
            out[o] += feats[i] @ W[k]          # each out row appears at most once per k
    return out


def batch_norm(x: np.ndarray, w, b, mean, var) -> np.ndarray:
    """nn.BatchNorm1d in eval mode (App. A.12): (x-mean) * rsqrt(var+eps) * w + b."""
    invstd = (F32(1.0) / np.sqrt(var.astype(F32) + F32(BN_EPS))).astype(F32)
    return ((x - mean.astype(F32)) * invstd * w.astype(F32) + b.astype(F32)).astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


# --------------------------------------------------------------------------------------
# parameter inventory (names as in the Lightning state_dict minus prefix)
# --------------------------------------------------------------------------------------
def layer_table(out_channels: int = 1):
    """Ordered list of (name, kind, K, C_in, C_out) for the 33 convolutions.
    kind in {conv5, down, up, conv3, lin}.  ``out_channels`` = width of `final` (1 for SPS and MapMOS,
    3 for 4DMOS: c_ws/src/mos4d/scripts/mos4d.py:15)."""
    t = [("conv0p1s1", "conv5", 125, 1, INIT_DIM)]
    enc_in = INIT_DIM
    downs = ["conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"]
    for i in range(4):
        t.append((downs[i], "down", 8, enc_in, enc_in))
        t += _block_entries(f"block{i + 1}", enc_in, PLANES[i])
        enc_in = PLANES[i]
    skip = [PLANES[2], PLANES[1], PLANES[0], INIT_DIM]
    ups = ["convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"]
    cur = PLANES[3]
    for i in range(4):
        t.append((ups[i], "up", 8, cur, PLANES[4 + i]))
        cin = PLANES[4 + i] + skip[i]
        t += _block_entries(f"block{5 + i}", cin, PLANES[4 + i])
        cur = PLANES[4 + i]
    t.append(("final", "lin", 1, PLANES[7], out_channels))
    return t


def _block_entries(name, cin, cout):
    e = [(f"{name}.0.conv1", "conv3", 81, cin, cout), (f"{name}.0.conv2", "conv3", 81, cout, cout)]
    if cin != cout:                                   # resnet.py:98
        e.append((f"{name}.0.downsample.0", "lin", 1, cin, cout))
    return e


def bn_table():
    """Ordered list of (name, C) of the 32 BatchNorm layers."""
    t = [("bn0", INIT_DIM)]
    enc = [INIT_DIM, PLANES[0], PLANES[1], PLANES[2]]
    for i in range(4):
        t.append((f"bn{i + 1}", enc[i]))
        t += _block_bn(f"block{i + 1}", enc[i], PLANES[i])
    for i in range(4):
        skip = [PLANES[2], PLANES[1], PLANES[0], INIT_DIM][i]
        t.append((f"bntr{4 + i}", PLANES[4 + i]))
        t += _block_bn(f"block{5 + i}", PLANES[4 + i] + skip, PLANES[4 + i])
    return t


def _block_bn(name, cin, cout):
    e = [(f"{name}.0.norm1", cout), (f"{name}.0.norm2", cout)]
    if cin != cout:
        e.append((f"{name}.0.downsample.1", cout))
    return e


def random_params(seed=0, randomize_bn=True, out_channels=1):
    """Synthetic weights: Kaiming-normal fan_out conv kernels
    (resnet.py:90; fan_out = K*C_out, std = sqrt(2/fan_out)), randomised BN statistics,
    final bias.  Keys follow the reference state_dict (App. B); 1x1 kernels are 2-D."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, kind, K, cin, cout in layer_table(out_channels):
        std = np.sqrt(2.0 / (K * cout))
        w = (rng.standard_normal((K, cin, cout)) * std).astype(F32)
        p[name + ".kernel"] = w[0] if kind == "lin" else w
    p["final.bias"] = (rng.standard_normal((1, out_channels)) * 0.1).astype(F32)
    for name, c in bn_table():
        if randomize_bn:
            p[name + ".bn.weight"] = rng.uniform(0.5, 1.5, c).astype(F32)
            p[name + ".bn.bias"] = (rng.standard_normal(c) * 0.1).astype(F32)
            p[name + ".bn.running_mean"] = (rng.standard_normal(c) * 0.1).astype(F32)
            p[name + ".bn.running_var"] = rng.uniform(0.5, 1.5, c).astype(F32)
        else:
            p[name + ".bn.weight"] = np.ones(c, F32)
            p[name + ".bn.bias"] = np.zeros(c, F32)
            p[name + ".bn.running_mean"] = np.zeros(c, F32)
            p[name + ".bn.running_var"] = np.ones(c, F32)
    return p


# --------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------
class CoordinateManager:
    """Minimal stand-in for ME's coordinate manager: the coordinate set per tensor
    stride and the kernel maps, built lazily and cached for one forward pass."""

    def __init__(self, coords_ts1: np.ndarray, cv=None):
        self.cv = cv
        self.coords = {1: coords_ts1}
        self.parent = {}       # ts -> row of each ts-voxel's parent in the 2ts set
        self._k3 = {}
        self._k5 = None
        self._kdown = {}

    def ensure_stride(self, ts2: int):
        if ts2 in self.coords:
            return
        ts = ts2 // 2
        self.ensure_stride(ts) if ts > 1 else None
        par = stride_coords(self.coords[ts], ts)
        uniq, inv = unique_first(par)
        self.coords[ts2] = uniq
        self.parent[ts] = inv

    def k3(self, ts):
        if ts not in self._k3:
            c = self.coords[ts]
            self._k3[ts] = kernel_map(c, c, kernel_offsets((3, 3, 3, 3), ts, self.cv))
        return self._k3[ts]

    def k5(self):
        if self._k5 is None:
            c = self.coords[1]
            self._k5 = kernel_map(c, c, kernel_offsets((5, 5, 5, 1), 1, self.cv))
        return self._k5

    def kdown(self, ts):
        """[2,2,2,1] stride-2 map from the ts set (in) to the 2ts set (out); also the
        map of the transposed conv 2ts -> ts with in/out swapped (App. A.9, A.10)."""
        if ts not in self._kdown:
            self.ensure_stride(2 * ts)
            self._kdown[ts] = kernel_map(self.coords[ts], self.coords[2 * ts],
                                         kernel_offsets((2, 2, 2, 1), ts, self.cv))
        return self._kdown[ts]


def _bn(p, name, x):
    return batch_norm(x, p[name + ".bn.weight"], p[name + ".bn.bias"],
                      p[name + ".bn.running_mean"], p[name + ".bn.running_var"])


def _basic_block(p, name, x, cm, ts):
    """BasicBlock (c_ws/src/mapmos/scripts/minkunet.py:65-82) with the optional
    1x1-conv+BN downsample of resnet.py:98-108."""
    n = len(x)
    y = sparse_conv(x, n, cm.k3(ts), p[name + ".conv1.kernel"])
    y = relu(_bn(p, name + ".norm1", y))
    y = sparse_conv(y, n, cm.k3(ts), p[name + ".conv2.kernel"])
    y = _bn(p, name + ".norm2", y)
    if (name + ".downsample.0.kernel") in p:
        cout = p[name + ".conv1.kernel"].shape[2]
        r = (x @ lin_kernel(p[name + ".downsample.0.kernel"], x.shape[1], cout, cm.cv)).astype(F32)
        r = _bn(p, name + ".downsample.1", r)
    else:
        r = x
    return relu(y + r)


def unet_forward(p, coords_ts1: np.ndarray, feats: np.ndarray, keep=False, cv=None):
    """MinkUNetBase.forward (minkunet.py:161-219) on CustomMinkUNet widths.
    Returns (logits [V1,1], CoordinateManager, intermediates dict if keep).  ``cv``: ME-convention options."""
    cm = CoordinateManager(coords_ts1, cv)
    mirrored = cv_get(cv, "transpose_index") == "mirrored"
    inter = {}

    def rec(name, v):
        if keep:
            inter[name] = v
        return v

    out = sparse_conv(feats, len(feats), cm.k5(), p["conv0p1s1.kernel"])
    out_p1 = rec("out_p1", relu(_bn(p, "bn0", out)))

    skips = {1: out_p1}
    cur = out_p1
    downs = ["conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"]
    ts = 1
    for i in range(4):
        km = cm.kdown(ts)
        nout = len(cm.coords[2 * ts])
        cur = sparse_conv(cur, nout, km, p[downs[i] + ".kernel"])
        cur = relu(_bn(p, f"bn{i + 1}", cur))
        ts *= 2
        cur = rec(f"block{i + 1}", _basic_block(p, f"block{i + 1}.0", cur, cm, ts))
        skips[ts] = cur

    ups = ["convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"]
    for i in range(4):
        fine = ts // 2
        km = cm.kdown(fine)
        Wup = p[ups[i] + ".kernel"]
        cur = sparse_conv(cur, len(cm.coords[fine]), km, Wup[::-1] if mirrored else Wup, transpose=True)
        cur = relu(_bn(p, f"bntr{4 + i}", cur))
        ts = fine
        cur = np.concatenate([cur, skips[ts]], axis=1)          # ME.cat(out, skip)
        cur = rec(f"block{5 + i}", _basic_block(p, f"block{5 + i}.0", cur, cm, ts))

    n_out = np.asarray(p["final.bias"]).size
    logits = (cur @ lin_kernel(p["final.kernel"], cur.shape[1], n_out, cv) + np.asarray(p["final.bias"]).reshape(1, -1)).astype(F32)
    return logits, cm, inter


def sigmoid(x):
    x = x.astype(F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def sps_forward(p, coordinates: np.ndarray, voxel_size: float, keep=False, cv=None):
    """SPSModel.forward (models.py:20-30): [N,5] float (b,x,y,z,t) -> scores [N].  ``cv``: ME-convention options."""
    q = quantize(coordinates, voxel_size)
    vox, inv = unique_first(q)
    feats = np.full((len(vox), 1), 0.5, dtype=F32)          # mean of 0.5's (App. A.4)
    logits, cm, inter = unet_forward(p, vox, feats, keep=keep, cv=cv)
    scores = sigmoid(logits[inv, 0])                         # slice + sigmoid (A.15)
    info = {"voxels": vox, "inverse": inv, "logits": logits[:, 0], "cm": cm, "inter": inter}
    return scores, info


def voxel_mean(features: np.ndarray, inverse: np.ndarray, n_vox: int) -> np.ndarray:
    """TensorField.sparse() with ME's default UNWEIGHTED_AVERAGE (App. A.4): voxel feature = arithmetic
    mean of its points' features; f32 accumulation in point order (CPU ME), then one division."""
    acc = np.zeros(n_vox, dtype=F32)
    cnt = np.zeros(n_vox, dtype=np.int64)
    np.add.at(acc, inverse, features.astype(F32).reshape(-1))
    np.add.at(cnt, inverse, 1)
    return (acc / np.maximum(cnt, 1).astype(F32)).astype(F32).reshape(-1, 1)


def head_forward(p, coordinates: np.ndarray, voxel_size: float, features=None, keep=False, cv=None):
    """The baseline heads on the same backbone: raw logits [N, out_channels] per point.
      * 4DMOS  (c_ws/src/mos4d/scripts/mos4d.py:17-32): constant 0.5 feature, 3-channel `final`, caller takes
        column 2;
      * MapMOS (c_ws/src/mapmos/scripts/mapmos.py:59-83): ``features`` [N] per point, voxel mean, 1 channel."""
    q = quantize(coordinates, voxel_size)
    vox, inv = unique_first(q)
    if features is None:
        feats = np.full((len(vox), 1), 0.5, dtype=F32)
    else:
        feats = voxel_mean(np.asarray(features), inv, len(vox))
    logits, cm, inter = unet_forward(p, vox, feats, keep=keep, cv=cv)
    info = {"voxels": vox, "inverse": inv, "logits": logits, "voxel_features": feats, "cm": cm, "inter": inter}
    return logits[inv], info


def mapmos_features(indices: np.ndarray) -> np.ndarray:
    """mapmos.py:65-71: 1 everywhere when all indices agree, else 1 + (i_max - i) / (i_max - i_min) in f32."""
    idx = np.asarray(indices, dtype=F32).reshape(-1)
    i_max, i_min = idx.max(), idx.min()
    if i_max == i_min:
        return np.ones_like(idx)
    return (F32(1) + (i_max - idx) / (i_max - i_min)).astype(F32)


# --------------------------------------------------------------------------------------
# variant-B submap (online path)
# --------------------------------------------------------------------------------------
def to_coords(cloud_xyz: np.ndarray, ds: float) -> np.ndarray:
    """util.to_coords_features (util.py:67-82): float32 division then ``.int()`` =
    truncation toward zero (App. A.2)."""
    q = np.array([ds, ds, ds], dtype=F32)
    return np.trunc(np.asarray(cloud_xyz, F32)[:, :3] / q).astype(np.int32)


def prune(map_coords: np.ndarray, scan_coords: np.ndarray, ds: float):
    """util.prune (util.py:85-114): unique map voxels INTERSECT unique scan voxels,
    returned as float32 voxel corners ``coords * ds`` and the unique-scan-voxel count.
    Row order of the reference (ME union/prune) is unspecified: set semantics."""
    su, _ = unique_first(scan_coords)
    mu, _ = unique_first(map_coords)
    idx = _CoordIndex(np.pad(mu, ((0, 0), (1, 1))), 1)
    hit = idx.lookup(np.pad(su, ((0, 0), (1, 1)))) >= 0
    inter = su[hit]
    # int32 tensor * python float -> float32 tensor in torch (util.py:112)
    return (inter.astype(F32) * F32(ds)).astype(F32), len(su)


# --------------------------------------------------------------------------------------
# metrics
# --------------------------------------------------------------------------------------
def calculate_metrics(true_labels: np.ndarray, predicted_labels: np.ndarray):
    """util.calculate_metrics (util.py:285-299), zero guards on P/R/F1 only."""
    tp = np.sum(np.logical_and(true_labels == 1, predicted_labels == 1))
    tn = np.sum(np.logical_and(true_labels == 0, predicted_labels == 0))
    fp = np.sum(np.logical_and(true_labels == 0, predicted_labels == 1))
    fn = np.sum(np.logical_and(true_labels == 1, predicted_labels == 0))
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = tp / (tp + fp) if (tp + fp) != 0 else 0
        recall = tp / (tp + fn) if (tp + fn) != 0 else 0
        f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) != 0 else 0
        accuracy = (tp + tn) / np.float64(tp + tn + fp + fn)
        diou = tp / np.float64(tp + fn + fp)
    return precision, recall, f1, accuracy, diou


def predict_metrics(scores: np.ndarray, batch: np.ndarray, epsilon: float):
    """SPSNet.predict_step (models.py:84-105) for one batch [N,6]=(b,x,y,z,t,label):
    MSE, R2 (torchmetrics R2Score), and calculate_metrics on eps-thresholded labels.
    Returns dict(loss, r2, precision, recall, f1, accuracy, dIoU)."""
    scan = batch[:, 4] == 1
    s = scores[scan].astype(np.float64)
    g = batch[scan, 5].astype(np.float64)
    loss = float(np.mean((s - g) ** 2)) if len(s) else float("nan")
    rss = np.sum((s - g) ** 2)
    tss = np.sum(g * g) - np.sum(g) * np.mean(g) if len(g) else 0.0
    r2 = float(1.0 - rss / tss) if tss != 0 else float("nan")
    pred = np.where(scores[scan] < F32(epsilon), 0, 1)
    gt = np.where(batch[scan, 5].astype(F32) < F32(epsilon), 0, 1)
    pr, rc, f1, acc, diou = calculate_metrics(gt, pred)
    return dict(loss=loss, r2=r2, precision=float(pr), recall=float(rc), f1=float(f1),
                accuracy=float(acc), dIoU=float(diou))
