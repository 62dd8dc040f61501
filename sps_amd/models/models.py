"""Host-side mirror of the reference's ``sps.models.models`` for the MI355X path.

  SPSModel(voxel_size).forward(coordinates[N,5]) -> scores[N]      reference models.py:13-30
  SPSNet(hparams, data_size=0, save_vis=False)                      reference models.py:33-111
      .forward(batch) .predict_step(batch, batch_idx) .load_state_dict(ckpt["state_dict"])
      .cuda() .eval() .freeze() ; list attributes predict_loss, predict_r2, dIoU, precision, recall, F1

Same names, argument meaning and state_dict keys; the arithmetic (quantise, voxel hash, kernel
maps, 33 sparse convs + BN, slice, sigmoid, metric sums) runs in hand-written HIP kernels
(sps_amd/csrc/sps_hip.hip) through the C ABI of include/sps_hip.h.  There is no CPU fallback:
a CPU tensor raises.  SPSNet is a plain nn.Module (pytorch_lightning is not a dependency);
training (models.py:62-82,154-160) is out of scope.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native
from .minkunet import CustomMinkUNet

_CONTEXTS: dict[tuple, "_native.Context"] = {}


def get_context(device_index: int, stream: int | None = None) -> "_native.Context":
    """One native context per (process, device, stream).

    A context is stream-ordered state (arena, hash tables, weights): work issued on different torch
    streams gets different contexts, so independent scans can be pipelined on several streams and their
    many small kernels overlap on the GPU.  ``stream`` is the raw hipStream_t handle (0 = default stream);
    None = the current torch stream of that device."""
    if stream is None:
        stream = torch.cuda.current_stream(device_index).cuda_stream if torch.cuda.is_available() else 0
    key = (device_index, int(stream))
    ctx = _CONTEXTS.get(key)
    if ctx is None:
        ctx = _CONTEXTS[key] = _native.Context(device_index)
    return ctx


def _require_device_tensor(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the MI355X (got device '{t.device}'): the sps_amd hot path "
                           "is HIP-only and has no CPU fallback")


class NativeBackboneModule(nn.Module):
    """Common part of the modules that own a ``self.MinkUNet`` parameter container and run it through
    libsps_hip.so: keeps the native weight blob of a (device, stream) context in sync with the parameters."""

    def _init_backbone(self, out_channels: int) -> None:
        self.MinkUNet = CustomMinkUNet(in_channels=1, out_channels=out_channels, D=4)
        self._dev_weights = {}           # device index -> _native.Weights holding this module's current parameters
        self._blob = None                # host copy of the weight blob in the native layout
        # any load_state_dict that reaches the backbone (predict.py:58 or util.py:39) re-uploads
        self.MinkUNet.register_load_state_dict_post_hook(lambda module, incompatible: self.mark_weights_dirty())

    # ---- weights -> native blob ---------------------------------------------------------
    def mark_weights_dirty(self) -> None:
        """Call after modifying parameters in place; load_state_dict / .cuda() / .to() do it themselves."""
        self._dev_weights = {}
        self._blob = None

    def _apply(self, fn, *args, **kwargs):
        self._dev_weights = {}
        self._blob = None
        return super()._apply(fn, *args, **kwargs)

    def device_weights(self, device_index: int) -> "_native.Weights":
        """The device-resident weight set of the current parameters: packed, permuted and uploaded ONCE per device and
        parameter version (sps_weights_create), then shared by every (device, stream) context."""
        w = self._dev_weights.get(device_index)
        if w is None:
            oc = self.MinkUNet.out_channels
            blob = self._blob
            if blob is None:
                sd = self.MinkUNet.state_dict()
                blob = np.empty(_native.lib.sps_head_numel(oc), dtype=np.float32)
                for name, off, numel in _native.weight_layout(oc):
                    t = sd[name].detach().to("cpu", torch.float32).contiguous().reshape(-1)
                    if t.numel() != numel:
                        raise ValueError(f"parameter {name} has {t.numel()} elements, the native layout expects {numel}")
                    blob[off: off + numel] = t.numpy()
                self._blob = blob
            w = self._dev_weights[device_index] = _native.Weights(device_index, blob.ctypes.data, blob.size, oc)
        return w

    def _sync_weights(self, ctx) -> None:
        w = self.device_weights(ctx.device)
        if ctx.weights is not w:         # a context is shared by every model on its device/stream
            ctx.set_weights(w)           # O(1), no synchronisation

    @staticmethod
    def _prepare_coordinates(coordinates: torch.Tensor) -> torch.Tensor:
        _require_device_tensor(coordinates, "coordinates")
        if coordinates.dim() != 2 or coordinates.shape[1] < 5:
            raise ValueError(f"coordinates must be [N, 5], got {tuple(coordinates.shape)}")
        if coordinates.dtype != torch.float32:
            coordinates = coordinates.to(torch.float32)
        if coordinates.stride(1) != 1:
            coordinates = coordinates.contiguous()
        return coordinates


class SPSModel(NativeBackboneModule):
    def __init__(self, voxel_size: float):
        super().__init__()
        self.voxel_size = float(voxel_size)
        # kept for API compatibility with models.py:16 (a plain attribute, not saved)
        self.quantization = torch.Tensor([1.0, voxel_size, voxel_size, voxel_size, 1.0])
        self._init_backbone(out_channels=1)
        self.sigmoid = nn.Sigmoid()

    # ---- forward --------------------------------------------------------------------------
    def forward(self, coordinates: torch.Tensor) -> torch.Tensor:
        """coordinates: float32 [N, >=5] rows (b, x, y, z, t); returns scores float32 [N]."""
        coordinates = self._prepare_coordinates(coordinates)
        with torch.cuda.device(coordinates.device):
            stream = torch.cuda.current_stream().cuda_stream
            ctx = get_context(coordinates.device.index or 0, stream)
            self._sync_weights(ctx)
            n = coordinates.shape[0]
            scores = torch.empty(n, dtype=torch.float32, device=coordinates.device)
            ctx.forward(coordinates.data_ptr(), coordinates.stride(0) if n else 5, n, self.voxel_size,
                        scores.data_ptr(), stream)
        return scores


class SPSNet(nn.Module):
    def __init__(self, hparams: dict, data_size=0, save_vis=False):
        super().__init__()
        self.hparams = hparams
        self.model = SPSModel(hparams["MODEL"]["VOXEL_SIZE"])
        self.save_vis = save_vis
        self.data_dir = str(os.environ.get("DATA"))
        self.test_seq = hparams["DATA"]["SPLIT"]["TEST"]
        self.epsilon = hparams["FILTER"]["THRESHOLD"]
        self.predict_loss = []
        self.predict_r2 = []
        self.dIoU = []
        self.precision = []
        self.recall = []
        self.F1 = []
        self.data_size = data_size

    def freeze(self):
        for p in self.parameters():
            p.requires_grad_(False)
        return self.eval()

    def forward(self, batch: torch.Tensor) -> torch.Tensor:
        return self.model(batch[:, :5])          # strided view: no copy crosses the boundary

    @torch.no_grad()
    def forward_metrics(self, batch: torch.Tensor, n_batches: int = 1, out: torch.Tensor | None = None):
        """forward + the per-batch-index metric sums of predict_step in ONE native call (sps_forward_metrics): returns
        (scores [N], sums float64 [n_batches, 8] on the device).  ``batch`` rows are (b,x,y,z,t,label); ``out`` may be
        a preallocated contiguous float64 device tensor of n_batches * 8 elements (e.g. a row of a results table)."""
        _require_device_tensor(batch, "batch")
        if batch.dim() != 2 or batch.shape[1] < 6:
            raise ValueError(f"batch must be [N, 6] = (b,x,y,z,t,label), got {tuple(batch.shape)}")
        if batch.dtype != torch.float32 or batch.stride(1) != 1:
            batch = batch.to(torch.float32).contiguous()
        with torch.cuda.device(batch.device):
            stream = torch.cuda.current_stream().cuda_stream
            ctx = get_context(batch.device.index or 0, stream)
            self.model._sync_weights(ctx)
            n = batch.shape[0]
            scores = torch.empty(n, dtype=torch.float32, device=batch.device)
            if out is None:
                out = torch.empty((n_batches, 8), dtype=torch.float64, device=batch.device)
            elif out.dtype != torch.float64 or not out.is_contiguous() or out.numel() != n_batches * 8 or not out.is_cuda:
                raise ValueError("out must be a contiguous float64 device tensor with n_batches * 8 elements")
            ctx.forward_metrics(batch.data_ptr(), batch.stride(0) if n else 6, n, self.model.voxel_size, float(self.epsilon),
                                n_batches, scores.data_ptr(), out.data_ptr(), stream)
        return scores, out

    @torch.no_grad()
    def step_metrics(self, batch: torch.Tensor, scores: torch.Tensor, n_batches: int = 1):
        """Per-batch-index sums [count, TP, FP, FN, TN, sum (s-g)^2, sum g, sum g^2] over scan rows
        (one device->host copy for the whole step; the reference does three, models.py:87,97-98)."""
        _require_device_tensor(batch, "batch")
        if batch.dtype != torch.float32 or batch.stride(1) != 1:
            batch = batch.to(torch.float32).contiguous()
        with torch.cuda.device(batch.device):
            stream = torch.cuda.current_stream().cuda_stream
            ctx = get_context(batch.device.index or 0, stream)
            return ctx.metrics(scores.data_ptr(), batch.data_ptr(), batch.stride(0), batch.shape[0],
                               float(self.epsilon), n_batches, stream)

    @torch.no_grad()
    def predict_step(self, batch: torch.Tensor, batch_idx: int, dataloader_idx: int = 0):
        """models.py:84-111: all scan rows (t == 1) of the batch tensor are pooled, exactly as
        the reference does (it runs with BATCH_SIZE forced to 1, predict.py:50)."""
        scores = self.forward(batch)
        nb = int(batch[:, 0].max().item()) + 1 if batch.shape[0] else 1
        sums = np.asarray(self.step_metrics(batch, scores, max(nb, 1)), dtype=np.float64).sum(axis=0)
        m = metrics_from_sums(sums)
        self.predict_loss.append(m["loss"])
        self.predict_r2.append(m["r2"])
        self.dIoU.append(m["dIoU"])
        self.precision.append(m["precision"])
        self.recall.append(m["recall"])
        self.F1.append(m["f1"])
        if self.save_vis:
            raise NotImplementedError("save_vis (.npy dumps, models.py:113-152) is out of scope")
        return m


def metrics_from_sums(s) -> dict:
    """[count, TP, FP, FN, TN, sse, sum g, sum g^2] -> MSE (nn.MSELoss), R2 (torchmetrics R2Score:
    1 - rss/tss, tss = sum g^2 - (sum g)^2/n) and util.calculate_metrics (util.py:285-299: zero
    guards on precision / recall / f1 only; accuracy and dIoU divide unguarded)."""
    n, tp, fp, fn, tn, sse, sg, sgg = (float(x) for x in s)
    nan = float("nan")
    loss = sse / n if n else nan
    tss = sgg - sg * sg / n if n else 0.0
    r2 = 1.0 - sse / tss if tss != 0 else nan
    precision = tp / (tp + fp) if (tp + fp) != 0 else 0
    recall = tp / (tp + fn) if (tp + fn) != 0 else 0
    f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) != 0 else 0
    accuracy = (tp + tn) / (tp + tn + fp + fn) if (tp + tn + fp + fn) != 0 else nan
    diou = tp / (tp + fn + fp) if (tp + fn + fp) != 0 else nan
    return dict(loss=loss, r2=r2, precision=precision, recall=recall, f1=f1, accuracy=accuracy, dIoU=diou,
                count=n, tp=tp, fp=fp, fn=fn, tn=tn)
