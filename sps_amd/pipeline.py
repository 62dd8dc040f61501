"""Streaming stable-points filter: the per-scan pipeline of the reference's online ROS node
(c_ws/src/sps_filter/scripts/sps_node.py:88-176) without the ROS transport:

    pose transform (:103)  ->  variant-B submap against the device-resident map hash (:111-115)
    ->  infer (:120)  ->  epsilon filter, keep score <= eps (:148)  ->  compacted filtered cloud
    + per-stage timers T / P / I like the node's log line (:164-176).

Stream-ordered end to end: four native calls (sps_filter_prepare, sps_forward_n, sps_compact_stable and the copy
of four counters) are issued back to back on the caller's stream without any host synchronisation -- the submap
size the forward depends on never visits the host (sps_forward_n reads its row count from device memory).
``submit()`` returns immediately; ``PendingFilter.result()`` is the one synchronisation (it also surfaces sticky
device errors such as out-of-range coordinates).  The map voxel hash is built once (the reference re-hashes the
whole map in every callback, util.py:86-89).
"""
from __future__ import annotations

import time
from dataclasses import dataclass

import numpy as np
import torch

from .models.models import _require_device_tensor, get_context


@dataclass
class FilterResult:
    filtered: torch.Tensor     # [m, 3] scan points (sensor frame, as received) with score <= eps
    scores: torch.Tensor       # [n] stability score of every scan point
    n_scan_voxels: int         # S in the node's log line
    n_submap_voxels: int       # M
    t_total: float             # seconds: host wall time submit -> result
    t_prune: float             # GPU seconds: transform + submap (hipEvents)
    t_infer: float             # GPU seconds: forward + epsilon filter (hipEvents)


class PendingFilter:
    """A scan whose work has been issued; everything lives on the device until result()."""

    def __init__(self, owner, n, raw32, batch, scores, filtered, counts, counts_host, events, stream, t0):
        self._o, self.n, self._raw32, self._batch, self._scores, self._filtered = owner, n, raw32, batch, scores, filtered
        self._counts, self._counts_host, self._ev, self._stream, self._t0 = counts, counts_host, events, stream, t0

    def result(self) -> FilterResult:
        self._stream.synchronize()                                   # the one host synchronisation of the scan
        self._o.ctx.check_errors(self._stream.cuda_stream)           # SPS_ERR_RANGE etc. (NaN scores are never silently dropped)
        n_sub, n_scan_vox, _, n_keep = (int(x) for x in self._counts_host.tolist())
        e0, e1, e2 = self._ev
        predicted_scan_labels = self._scores[: self.n]
        # sps_node.py:147 (same message): one score per scan point as received
        assert len(predicted_scan_labels) == self.n, \
            f"Predicted scans labels len ({len(predicted_scan_labels)}) does not equal scan len ({self.n})"
        assert 0 <= n_keep <= self.n, f"filtered scan len ({n_keep}) exceeds scan len ({self.n})"
        return FilterResult(self._filtered[:n_keep], predicted_scan_labels, n_scan_vox, n_sub,
                            time.time() - self._t0, e0.elapsed_time(e1) * 1e-3, e1.elapsed_time(e2) * 1e-3)


class StableFilter:
    def __init__(self, model, map_points, voxel_size: float = 0.1, epsilon: float = 0.84, device="cuda"):
        self.model, self.ds, self.epsilon = model, float(voxel_size), float(epsilon)
        self.device = torch.device(device if torch.device(device).index is not None else f"cuda:{torch.cuda.current_device()}")
        self.map_xyz = torch.as_tensor(map_points)[:, :3].to(torch.float32).to(self.device).contiguous()   # sps_node.py:69-74
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream()
            self.ctx = get_context(self.device.index, self.stream.cuda_stream)
            self.ctx.map_upload(self.map_xyz.data_ptr(), 3, len(self.map_xyz), self.ds, self.stream.cuda_stream)

    @torch.no_grad()
    def submit(self, scan_xyz, pose=None) -> PendingFilter:
        t0 = time.time()
        raw = torch.as_tensor(scan_xyz)
        if raw.dtype not in (torch.float32, torch.float64):
            raw = raw.to(torch.float32)
        raw = raw.to(self.device, non_blocking=True)
        if raw.dim() != 2 or raw.shape[1] < 3:
            raise ValueError(f"scan must be [n, >=3], got {tuple(raw.shape)}")
        if raw.stride(1) != 1:
            raw = raw.contiguous()
        n = raw.shape[0]
        raw32 = raw if raw.dtype == torch.float32 else raw[:, :3].to(torch.float32)
        T = None if pose is None else np.asarray(pose, dtype=np.float64)
        if T is not None and T.shape != (4, 4):
            raise ValueError("pose must be a 4x4 matrix")
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream()
            if st.cuda_stream != self.stream.cuda_stream:
                raise RuntimeError("StableFilter must be called on the stream it was created on (its map hash lives in that "
                                   "stream's native context)")
            self.model.model._sync_weights(self.ctx)
            batch = torch.empty((2 * max(n, 1), 5), dtype=torch.float32, device=self.device)
            scores = torch.empty(2 * max(n, 1), dtype=torch.float32, device=self.device)
            filtered = torch.empty((max(n, 1), 3), dtype=torch.float32, device=self.device)
            counts = torch.zeros(4, dtype=torch.int32, device=self.device)
            counts_host = torch.empty(4, dtype=torch.int32).pin_memory()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            s = st.cuda_stream
            ev[0].record(st)
            # :103 transform, :111-115 submap, util.py:166-176 tensor assembly -- rows and counts stay on the device
            self.ctx.filter_prepare(raw.data_ptr(), raw.dtype == torch.float64, raw.stride(0), n, T, batch.data_ptr(),
                                    counts.data_ptr(), s)
            ev[1].record(st)
            # :120 infer: the row count n + n_sub is read from counts[2] on the device
            self.ctx.forward_n(batch.data_ptr(), 5, 2 * n, counts.data_ptr() + 8, float(self.model.model.voxel_size),
                               scores.data_ptr(), s)
            # :147-148 epsilon filter on the points as received
            self.ctx.compact_stable(scores.data_ptr(), raw32.data_ptr(), raw32.stride(0), 3, n, self.epsilon,
                                    filtered.data_ptr(), counts.data_ptr() + 12, s)
            ev[2].record(st)
            counts_host.copy_(counts, non_blocking=True)
        return PendingFilter(self, n, raw32, batch, scores, filtered, counts, counts_host, ev, st, t0)

    def __call__(self, scan_xyz, pose=None) -> FilterResult:
        return self.submit(scan_xyz, pose).result()
