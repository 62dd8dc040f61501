"""Host-side mirror of the reference's ``sps.models.models`` for the MI355X path.

  SPSModel(voxel_size).forward(coordinates[N,5]) -> scores[N]      reference models.py:13-30
  SPSNet(hparams, data_size=0, save_vis=False)                      reference models.py:33-111
      .forward(batch) .predict_step(batch, batch_idx) .load_state_dict(ckpt["state_dict"])
      .cuda() .eval() .freeze() ; list attributes predict_loss, predict_r2, dIoU, precision, recall, F1

Same names, argument meaning and state_dict keys; the arithmetic (quantise, voxel hash, kernel
maps, 33 sparse convs + BN, slice, sigmoid, metric sums) runs in hand-written HIP kernels
(sps_amd/csrc/sps_hip.hip) through the C ABI of include/sps_hip.h.  There is no CPU fallback:
a CPU tensor raises.  SPSNet is a plain nn.Module (pytorch_lightning is not a dependency) with the
LightningModule hooks the reference defines (forward, common_step, training_step, validation_step, predict_step,
configure_optimizers); in ``.train()`` mode the forward is differentiable (train-mode BatchNorm, native backward).
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native, conventions
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

    def _init_backbone(self, out_channels: int, me_conventions=None) -> None:
        self.MinkUNet = CustomMinkUNet(in_channels=1, out_channels=out_channels, D=4)
        self._dev_weights = {}           # device index -> _native.Weights holding this module's current parameters
        self._blob = None                # host copy of the weight blob in the native layout
        # which MinkowskiEngine conventions the parameters are written in (sps_amd/conventions.py: the ONE switch-point);
        # the library computes in the canonical one, the blob is permuted where it is packed
        self.me_conventions = conventions.parse(me_conventions)
        self._blob_perm = {}
        # any load_state_dict that reaches the backbone (predict.py:58 or util.py:39) re-uploads
        self.MinkUNet.register_load_state_dict_post_hook(lambda module, incompatible: self._on_load_state_dict())
        self.MinkUNet._register_load_state_dict_pre_hook(self._accept_lin_kernel_shapes, with_module=True)

    def set_me_conventions(self, spec) -> None:
        """Declare the ME conventions the current / future parameters follow (None = conventions.DEFAULT)."""
        self.me_conventions = conventions.parse(spec)
        self.mark_weights_dirty()
        self._plan = None

    def _accept_lin_kernel_shapes(self, module, state_dict, prefix, *_):
        """A ``kernel_size = 1`` kernel may arrive 3-D ``[1, C_in, C_out]`` (always accepted: same memory as the 2-D form) or,
        ONLY when ``me_conventions.lin_layout == "out_in"`` has been declared, as ``[C_out, C_in]``: the stored MEMORY is kept,
        what it means is decided where the blob is packed.  Any other shape still fails the strict load."""
        out_in = self.me_conventions.lin_layout == "out_in"
        for name, p in module.named_parameters():
            key = prefix + name
            if name.endswith(".kernel") and p.dim() == 2 and key in state_dict:
                t = state_dict[key]
                squeezed = t.dim() == 3 and t.shape[0] == 1 and tuple(t.shape[1:]) == tuple(p.shape)
                swapped = out_in and t.dim() == 2 and tuple(t.shape) == tuple(p.shape)[::-1]
                if t.shape != p.shape and (squeezed or swapped):
                    state_dict[key] = t.reshape(p.shape)

    def blob_permutation(self):
        """perm with ``blob_internal = blob_as_stored[perm]`` (host int64 numpy) or None for the canonical conventions."""
        cv = self.me_conventions
        if cv.is_default:
            return None
        oc = self.MinkUNet.out_channels
        perm = self._blob_perm.get((cv, oc))
        if perm is None:
            sd = self.MinkUNet.state_dict()
            shapes = {}
            for name, _, _ in _native.weight_layout(oc):
                if name.endswith(".kernel"):
                    sh = tuple(sd[name].shape)
                    shapes[name] = sh if len(sh) == 3 else (1,) + sh
            perm = self._blob_perm[(cv, oc)] = conventions.blob_permutation(_native.weight_layout(oc), shapes, cv)
        return perm

    # ---- weights -> native blob ---------------------------------------------------------
    def _on_load_state_dict(self) -> None:
        self.mark_weights_dirty()
        self._plan = None                # load_state_dict(assign=True) replaces the tensors: the training plan is rebuilt

    def mark_weights_dirty(self) -> None:
        """Call after modifying parameters in place; load_state_dict / .cuda() / .to() do it themselves."""
        self._dev_weights = {}
        self._blob = None

    def _apply(self, fn, *args, **kwargs):
        self._dev_weights = {}
        self._blob = None
        self._plan = None                # .to() / .cuda() replace the buffer tensors: the training plan is rebuilt
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
                perm = self.blob_permutation()
                if perm is not None:     # the checkpoint's conventions -> the library's (conventions.py)
                    blob = np.ascontiguousarray(blob[perm])
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


def _bn_of_conv(conv: str) -> str:
    """state_dict prefix of the BatchNorm that follows conv `conv` (minkunet.py:55-159, resnet BasicBlock)."""
    if conv == "conv0p1s1":
        return "bn0"
    if conv.startswith("convtr"):
        return "bntr" + conv[len("convtr")]
    if conv.startswith("conv") and conv[4].isdigit():
        return "bn" + conv[4]
    return conv.replace("conv1", "norm1").replace("conv2", "norm2").replace("downsample.0", "downsample.1")


class _TrainForward(torch.autograd.Function):
    """Train-mode forward / backward of the SPS network through libsps_hip.so (sps_train_forward / _backward): the
    autograd node that stands where the reference has MinkowskiEngine's autograd functions (models.py:62-76)."""

    sync_gradients = True     # average the flat gradient over the ranks of an initialised process group

    @staticmethod
    def forward(fctx, module, coordinates, *params):
        dev = coordinates.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            ctx = get_context(dev.index or 0, stream)
            plan = module._train_plan(dev)          # parameters + BN buffers: views of one tensor in the native layout
            # non-canonical ME conventions: the library sees the permuted copy, the gradient goes back through the inverse
            blob = plan.flat if plan.perm is None else plan.flat.index_select(0, plan.perm)
            n = coordinates.shape[0]
            scores = torch.empty(n, dtype=torch.float32, device=dev)
            stats = torch.zeros(plan.n_stats, dtype=torch.float32, device=dev)
            ctx.train_forward(blob.data_ptr(), blob.numel(), coordinates.data_ptr(), coordinates.stride(0), n,
                              module.voxel_size, scores.data_ptr(), stats.data_ptr(), stream)
            generation = ctx.train_generation()     # the activations this node's backward needs live in the context
        fctx.save_for_backward(scores)
        fctx.native = (ctx, [plan.span_of[id(p)] + (tuple(p.shape),) for p in params], blob.numel(), generation, plan.inv_perm)
        fctx.mark_non_differentiable(stats)
        return scores, stats

    @staticmethod
    def backward(fctx, dscores, _dstats):
        (scores,) = fctx.saved_tensors
        ctx, spans, numel, generation, inv_perm = fctx.native
        with torch.cuda.device(scores.device):
            stream = torch.cuda.current_stream().cuda_stream
            grad = torch.empty(numel, dtype=torch.float32, device=scores.device)
            d = dscores.to(torch.float32).contiguous()
            # fails (SpsError) if a later forward on this context overwrote the activations: no silently wrong gradients
            ctx.train_backward(d.data_ptr(), scores.data_ptr(), grad.data_ptr(), numel, stream, generation)
            if inv_perm is not None:
                grad = grad.index_select(0, inv_perm)
            # data-parallel training (one process per GPU, scripts/train.py under torchrun): the gradient of the whole
            # network is ONE flat tensor, so the ranks exchange it with a single all-reduce (RCCL over xGMI, 7.4 MB)
            # instead of one per parameter
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and _TrainForward.sync_gradients:
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)
                grad.div_(dist.get_world_size())
        out = [grad[off: off + num].view(shape) if need else None
               for (off, num, shape), need in zip(spans, fctx.needs_input_grad[2:])]
        return (None, None, *out)


class _ScanMSE(torch.autograd.Function):
    """nn.MSELoss over the rows with t == 1 and the R2Score of the same rows (models.py:62-72) in two launches
    (sps_scan_mse), the gradient wrt the scores in one (sps_scan_mse_backward).  ``batch``: float32 rows (b,x,y,z,t,label)."""

    @staticmethod
    def forward(fctx, scores, batch):
        dev = scores.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            work = torch.empty(4 * 257, dtype=torch.float64, device=dev)
            out = torch.empty(2, dtype=torch.float32, device=dev)
            ld = batch.stride(0)
            _native.check(_native.lib.sps_scan_mse(scores.data_ptr(), batch.data_ptr() + 20, ld, batch.data_ptr() + 16, ld,
                                                   scores.shape[0], work.data_ptr(), out.data_ptr(), stream))
        fctx.save_for_backward(scores, batch, work)
        loss, r2 = out[0], out[1]
        fctx.mark_non_differentiable(r2)
        return loss, r2

    @staticmethod
    def backward(fctx, gloss, _gr2):
        scores, batch, work = fctx.saved_tensors
        with torch.cuda.device(scores.device):
            stream = torch.cuda.current_stream().cuda_stream
            g = gloss.to(torch.float32).contiguous()
            dscores = torch.empty_like(scores)
            ld = batch.stride(0)
            _native.check(_native.lib.sps_scan_mse_backward(scores.data_ptr(), batch.data_ptr() + 20, ld, batch.data_ptr() + 16, ld,
                                                            scores.shape[0], work.data_ptr(), g.data_ptr(), dscores.data_ptr(), stream))
        return dscores, None


class _TrainPlan:
    """What a training step needs from the module, computed once per placement of the module's tensors."""
    __slots__ = ("flat", "checks", "params", "span_of", "n_stats", "run_idx", "stat_idx", "nbt", "perm", "inv_perm")


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
        """coordinates: float32 [N, >=5] rows (b, x, y, z, t); returns scores float32 [N].
        In training mode (``.train()``, gradients enabled) the forward runs train-mode BatchNorm and is differentiable
        with respect to every parameter (sps_train_forward / sps_train_backward); otherwise the inference path."""
        coordinates = self._prepare_coordinates(coordinates)
        if self.training and torch.is_grad_enabled():
            return self._train_forward(coordinates)
        with torch.cuda.device(coordinates.device):
            stream = torch.cuda.current_stream().cuda_stream
            ctx = get_context(coordinates.device.index or 0, stream)
            self._sync_weights(ctx)
            n = coordinates.shape[0]
            scores = torch.empty(n, dtype=torch.float32, device=coordinates.device)
            ctx.forward(coordinates.data_ptr(), coordinates.stride(0) if n else 5, n, self.voxel_size,
                        scores.data_ptr(), stream)
        return scores

    def _train_forward(self, coordinates: torch.Tensor) -> torch.Tensor:
        if coordinates.shape[0] == 0:
            raise ValueError("a training step needs at least one point")
        plan = self._train_plan(coordinates.device)
        scores, stats = _TrainForward.apply(self, coordinates, *plan.params)
        self._update_running_stats(plan, stats)
        self.mark_weights_dirty()          # the optimiser is about to change the parameters: eval contexts re-upload
        return scores

    @torch.no_grad()
    def _train_plan(self, device) -> _TrainPlan:
        """Every parameter and BatchNorm buffer of the backbone as a VIEW of one flat float32 device tensor in the native
        blob layout (sps_weights_tensor_info): the training step hands the library one pointer instead of concatenating
        194 tensors, and the optimiser's in-place updates keep it current.  Rebuilt when the storage moved (.cuda(), .to(),
        a parameter's .data re-assigned): every step checks the 194 addresses, nothing else is recomputed."""
        plan = getattr(self, "_plan", None)
        if plan is not None and plan.flat.device == device:
            base = plan.flat.data_ptr()
            if all(t.data_ptr() == base + byte_off for t, byte_off in plan.checks):
                return plan
        layout = _native.weight_layout(self.MinkUNet.out_channels)
        sd = self.MinkUNet.state_dict(keep_vars=True)
        flat = torch.cat([sd[name].detach().reshape(-1).to(device=device, dtype=torch.float32) for name, _, _ in layout])
        for name, off, num in layout:
            t = sd[name]
            t.data = flat[off: off + num].view(t.shape)
        plan = _TrainPlan()
        plan.flat = flat
        plan.checks = [(sd[name], 4 * off) for name, off, _ in layout]
        plan.params = list(self.MinkUNet.parameters())
        plan.span_of = {id(sd[name]): (off, num) for name, off, num in layout}
        # BatchNorm running statistics: their positions in the flat tensor / of their batch values in the `stats` vector the
        # library returns (per conv in layout order: mean, biased var, unbiased var, C values each)
        span = {name: (off, num) for name, off, num in layout}
        run, stat, nbt, soff = [], [], [], 0
        for name, _, _ in layout:
            if not name.endswith(".kernel"):
                continue
            conv = name[: -len(".kernel")]
            c = int(sd[name].shape[-1])
            if conv != "final":
                bn = _bn_of_conv(conv)
                for which, col in ((".bn.running_mean", 0), (".bn.running_var", 2)):
                    off, num = span[bn + which]
                    assert num == c
                    run.append(torch.arange(off, off + c))
                    stat.append(torch.arange(soff + col * c, soff + (col + 1) * c))
                nbt.append(sd[bn + ".bn.num_batches_tracked"])
            soff += 3 * c
        plan.n_stats = soff
        plan.run_idx = torch.cat(run).to(device)
        plan.stat_idx = torch.cat(stat).to(device)
        plan.nbt = nbt
        perm = self.blob_permutation()
        plan.perm = None if perm is None else torch.from_numpy(perm).to(device)
        plan.inv_perm = None if perm is None else torch.from_numpy(conventions.inverse_permutation(perm)).to(device)
        self._plan = plan
        return plan

    @torch.no_grad()
    def _update_running_stats(self, plan: _TrainPlan, stats: torch.Tensor) -> None:
        """nn.BatchNorm1d bookkeeping in train mode: running = (1 - m) running + m batch (m = 0.1; the library delivers
        the batch mean and the UNBIASED batch variance), num_batches_tracked += 1 -- six small kernels on index vectors
        that were built once, no synchronisation."""
        m = 0.1
        cur = plan.flat.index_select(0, plan.run_idx)
        cur.mul_(1 - m).add_(stats.index_select(0, plan.stat_idx), alpha=m)
        plan.flat.index_copy_(0, plan.run_idx, cur)
        torch._foreach_add_(plan.nbt, 1)


class SPSNet(nn.Module):
    def __init__(self, hparams: dict, data_size=0, save_vis=False):
        super().__init__()
        self.hparams = hparams
        self.model = SPSModel(hparams["MODEL"]["VOXEL_SIZE"])
        # optional, not in the reference's config.yaml: MODEL.ME_CONVENTIONS = {option: value} or "option=value,..."
        # (sps_amd/conventions.py) for a checkpoint whose MinkowskiEngine build indexes its kernels differently
        if hparams["MODEL"].get("ME_CONVENTIONS"):
            self.model.set_me_conventions(hparams["MODEL"]["ME_CONVENTIONS"])
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
        self.loss = nn.MSELoss()                 # models.py:52

    def freeze(self):
        for p in self.parameters():
            p.requires_grad_(False)
        return self.eval()

    def forward(self, batch: torch.Tensor) -> torch.Tensor:
        return self.model(batch[:, :5])          # strided view: no copy crosses the boundary

    # ---- training (models.py:62-82, :154-160) -------------------------------------------------------------------
    @staticmethod
    def r2score(preds: torch.Tensor, target: torch.Tensor, weight: torch.Tensor = None, count=None) -> torch.Tensor:
        """torchmetrics.R2Score: 1 - sum (y - p)^2 / sum (y - mean y)^2 (over the points of weight 1 when weights are given)."""
        if weight is None:
            weight = torch.ones_like(target)
            count = target.numel()
        ss_res = torch.sum(weight * (target - preds) ** 2)
        mean = torch.sum(weight * target) / count
        ss_tot = torch.sum(weight * (target - mean) ** 2)
        return 1.0 - ss_res / ss_tot

    def common_step(self, batch: torch.Tensor):
        coordinates = batch[:, :5].reshape(-1, 5)
        gt_labels = batch[:, 5].reshape(-1)
        scores = self.model(coordinates)
        # the reference selects the scan's points (t == 1) with np.where on the host (models.py:65-68): here the same
        # means are taken over the rows with t == 1 on the device, so the step never waits for the GPU (no index list, no
        # size to learn) -- in the library's two launches when the batch is a plain float32 matrix on the GPU
        if (scores.is_cuda and batch.is_cuda and batch.dtype == torch.float32 and batch.dim() == 2 and batch.shape[1] >= 6
                and batch.stride(1) == 1 and scores.dtype == torch.float32 and scores.is_contiguous()
                and scores.shape[0] > 0):     # (an empty batch takes the torch formulation below: NaN loss, as the reference's)
            return _ScanMSE.apply(scores, batch)
        w = (coordinates[:, 4] == 1).to(scores.dtype)
        cnt = w.sum()
        loss = (w * (scores - gt_labels) ** 2).sum() / cnt            # nn.MSELoss over the selected points
        r2 = self.r2score(scores.detach(), gt_labels, w, cnt)
        return loss, r2

    def training_step(self, batch: torch.Tensor, batch_idx: int, dataloader_idx: int = 0):
        loss, r2 = self.common_step(batch)
        return {"loss": loss, "val_r2": r2}

    def validation_step(self, batch: torch.Tensor, batch_idx: int):
        loss, r2 = self.common_step(batch)
        return {"val_loss": loss, "val_r2": r2}

    def configure_optimizers(self):
        params = list(self.parameters())
        # same Adam as the reference (models.py:154-160); on the GPU the single-launch ("fused") implementation of it
        fused = bool(params) and all(p.is_cuda for p in params) and os.environ.get("SPS_FUSED_ADAM", "1") != "0"
        optimizer = torch.optim.Adam(params, lr=self.hparams["TRAIN"]["LR"],
                                     weight_decay=self.hparams["TRAIN"]["WEIGHT_DECAY"], fused=fused)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=self.hparams["TRAIN"]["LR_EPOCH"],
                                                    gamma=self.hparams["TRAIN"]["LR_DECAY"])
        return [optimizer], [scheduler]

    @torch.no_grad()
    def forward_metrics(self, batch: torch.Tensor, n_batches: int = 1, out: torch.Tensor | None = None,
                        scores: torch.Tensor | None = None, ctx=None):
        """forward + the per-batch-index metric sums of predict_step in ONE native call (sps_forward_metrics): returns
        (scores [N], sums float64 [n_batches, 8] on the device).  ``batch`` rows are (b,x,y,z,t,label); ``out`` may be
        a preallocated contiguous float64 device tensor of n_batches * 8 elements (e.g. a row of a results table);
        ``ctx``: the native context to run on (default: the shared context of the current stream)."""
        _require_device_tensor(batch, "batch")
        if batch.dim() != 2 or batch.shape[1] < 6:
            raise ValueError(f"batch must be [N, 6] = (b,x,y,z,t,label), got {tuple(batch.shape)}")
        if batch.dtype != torch.float32 or batch.stride(1) != 1:
            batch = batch.to(torch.float32).contiguous()
        with torch.cuda.device(batch.device):
            stream = torch.cuda.current_stream().cuda_stream
            if ctx is None:
                ctx = get_context(batch.device.index or 0, stream)
            self.model._sync_weights(ctx)
            n = batch.shape[0]
            if scores is None:
                scores = torch.empty(n, dtype=torch.float32, device=batch.device)
            elif scores.dtype != torch.float32 or not scores.is_contiguous() or scores.numel() < n or not scores.is_cuda:
                raise ValueError("scores must be a contiguous float32 device tensor with at least N elements")
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
            save_vis(os.path.join(self.data_dir, "predictions", self.test_seq[0]), batch, batch_idx, scores)
        return m


def save_vis(root: str, batch: torch.Tensor, batch_idx: int, scores: torch.Tensor) -> list:
    """The .npy dumps of models.py:113-152: per batch index b, ``<root>/scans/<batch_idx>_<b>.npy`` = scan rows
    (x, y, z, label, score) and ``<root>/maps/<batch_idx>_<b>.npy`` = submap rows (x, y, z, label).  As in the
    reference the score column is the POOLED scan scores of the whole batch tensor, so a batch of more than one scan
    trips the same length assertion (prediction runs with BATCH_SIZE 1, predict.py:50).  Returns the written paths."""
    s_path, m_path = os.path.join(root, "scans"), os.path.join(root, "maps")
    os.makedirs(s_path, exist_ok=True)
    os.makedirs(m_path, exist_ok=True)
    rows = batch.detach().cpu().numpy()
    sc = scores.detach().cpu().numpy().reshape(-1)
    is_scan = rows[:, -2] == 1
    pooled = sc[is_scan]
    written = []
    for b in np.unique(rows[:, 0]):
        of_b = rows[:, 0] == b
        scan, submap = rows[of_b & is_scan], rows[of_b & ~is_scan & (rows[:, -2] == 0)]
        assert len(scan) == len(pooled), "Lengths of arrays are not equal."
        b_name = str(b.item())                                                 # a float column: '0.0', as torch's .item()
        scan_pth = os.path.join(s_path, f"{batch_idx}_{b_name}.npy")
        map_pth = os.path.join(m_path, f"{batch_idx}_{b_name}.npy")
        np.save(scan_pth, np.column_stack((scan[:, 1:4], scan[:, -1], pooled)))
        np.save(map_pth, np.column_stack((submap[:, 1:4], submap[:, -1])))
        written += [scan_pth, map_pth]
    return written


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
