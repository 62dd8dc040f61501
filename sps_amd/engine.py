"""Pipelined per-scan evaluation loop: the reference's ``trainer.predict(model, dataloader)`` loop
(scripts/predict.py:64-67 driving SPSNet.predict_step, models.py:84-111) re-designed for one MI355X.

A forward is ~30 short kernels, most of them too small to fill 256 CUs on their own, so independent scans are kept
in flight on ``streams`` HIP streams (one native context each; all contexts share ONE device-resident weight set).
Per scan the engine issues, stream-ordered and without any host synchronisation:

    [optional H2D copy of the [N,6] batch from a pinned host buffer]  ->  sps_forward_metrics
    (quantise + voxel hash + pyramid + kernel maps + 33 sparse convs + slice + sigmoid + metric sums)

and the 8 metric sums of every batch index land in a row of a device-resident table.  ``finish()`` is the single
synchronisation point of a sequence: it drains the streams, surfaces sticky device errors (SPS_ERR_RANGE) and
returns the table.  bench.py, scripts/predict.py and the streaming filter all run THIS loop, so the measured path
is the product path.
"""
from __future__ import annotations

import numpy as np
import torch

from .models.models import SPSNet, _require_device_tensor, get_context, metrics_from_sums

DEFAULT_STREAMS = 23      # the HIP runtime multiplexes streams onto 4 hardware queues; 4k+3 maps best


class ScanEngine:
    def __init__(self, net: SPSNet, device: torch.device | int | None = None, streams: int = DEFAULT_STREAMS,
                 max_rows: int = 0, table_rows: int = 0, stage_cols: int = 0, compact: bool = True):
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.index = self.device.index or 0
        self.net = net
        self.eps = float(net.epsilon)
        self.vs = float(net.model.voxel_size)
        S = max(1, int(streams))
        with torch.cuda.device(self.device):
            self.main = torch.cuda.current_stream()
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(S)] if S > 1 else [self.main]
            self.ctxs = [get_context(self.index, st.cuda_stream) for st in self.streams]
        # LiDAR-sized arenas (2.2 KB instead of 6.9 KB of device memory per point and context): a cloud whose coarse
        # levels do not thin out like a LiDAR scan's aborts its forward on the device; finish() then raises SpsError
        # (SPS_ERR_NOMEM) with the contexts already switched to full-size arenas, and run_sequence() re-runs once
        self.compact = bool(compact)
        if self.compact:
            for cx in self.ctxs:
                cx.set_level_fractions(cx.LIDAR_FRACTIONS)
        # the engine only runs inference: the rulebook replaces the neighbour table where the layers run pair-exact
        for cx in self.ctxs:
            cx.set_inference_only(True)
        self._next = 0
        self._stage = [None] * S          # per-stream device staging buffer for host batches
        self._pinned = [None] * S
        self.table = None
        self.rows_used = 0
        self.prepare(max_rows, table_rows, stage_cols)

    # ---- set-up: everything that is not steady state happens here ------------------------------------------------
    def prepare(self, max_rows: int = 0, table_rows: int = 0, stage_cols: int = 0) -> None:
        """Arena of every context sized for ``max_rows`` points, the shared weight set attached, one tiny forward per
        context (first-use costs: code objects, events), and -- when batches will arrive as host tensors with
        ``stage_cols`` columns -- the per-stream device staging buffers: the first timed scan is already steady state."""
        with torch.cuda.device(self.device):
            w = self.net.model.device_weights(self.index)
            warm = torch.zeros((64, 6), dtype=torch.float32, device=self.device)
            warm[:, 1] = torch.arange(64, device=self.device) * 0.05
            warm[:, 4] = 1.0
            out = torch.zeros((1, 8), dtype=torch.float64, device=self.device)
            for cx, st in zip(self.ctxs, self.streams):
                if max_rows:
                    cx.reserve(int(max_rows))
                if cx.weights is not w:
                    cx.set_weights(w)
                with torch.cuda.stream(st):
                    self.net.forward_metrics(warm, 1, out)
            if table_rows:
                self.reset_table(table_rows)
            if stage_cols and max_rows:
                for k, st in enumerate(self.streams):
                    with torch.cuda.stream(st):
                        self._stage[k] = torch.empty((int(max_rows), int(stage_cols)), dtype=torch.float32, device=self.device)
            torch.cuda.synchronize(self.device)

    def reset_table(self, rows: int) -> None:
        if self.table is None or self.table.shape[0] < rows:
            self.table = torch.zeros((int(rows), 8), dtype=torch.float64, device=self.device)
        self.rows_used = 0

    # ---- steady state ----------------------------------------------------------------------------------------------
    def submit(self, batch: torch.Tensor, n_batches: int = 1, row: int | None = None) -> torch.Tensor:
        """One step: ``batch`` [N,6] = (b,x,y,z,t,label) rows with b < n_batches (BacchusModule.collate_fn layout).
        A host tensor is copied to the device on the scan's stream (pinned staging, asynchronous).  The metric sums of
        batch index b go to table row ``row + b`` (default: the next free rows).  Returns the scores [N] (device tensor,
        valid after finish() or a wait on stream ``self.last_stream``)."""
        k = self._next
        self._next = (k + 1) % len(self.streams)
        st = self.streams[k]
        if row is None:
            row = self.rows_used
        if self.table is None or row + n_batches > self.table.shape[0]:
            raise ValueError("metric table too small: call reset_table(rows) with the number of scans of the sequence")
        self.rows_used = max(self.rows_used, row + n_batches)
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            if not batch.is_cuda:
                batch = self._to_device(k, batch)
            scores, _ = self.net.forward_metrics(batch, n_batches, self.table[row: row + n_batches])
        self.last_stream = st
        return scores

    def _to_device(self, k: int, host: torch.Tensor) -> torch.Tensor:
        n = host.shape[0]
        if host.dtype != torch.float32 or not host.is_contiguous():
            host = host.to(torch.float32).contiguous()
        buf = self._stage[k]
        if buf is None or buf.shape[0] < n or buf.shape[1] != host.shape[1]:
            cap = max(n, 1024)
            buf = self._stage[k] = torch.empty((cap + cap // 4, host.shape[1]), dtype=torch.float32, device=self.device)
        if not host.is_pinned():
            if self._pinned[k] is None or self._pinned[k].shape[0] < n or self._pinned[k].shape[1] != host.shape[1]:
                self._pinned[k] = torch.empty((max(n, buf.shape[0]), host.shape[1]), dtype=torch.float32).pin_memory()
            # the stream's previous copy out of this pinned buffer must be done before it is overwritten
            self.streams[k].synchronize()
            self._pinned[k][:n].copy_(host)
            host = self._pinned[k][:n]
        dst = buf[:n]
        dst.copy_(host, non_blocking=True)
        return dst

    def use_full_arenas(self) -> None:
        """Every context back to full-size arenas (cannot overflow; re-allocated by the next forward)."""
        self.compact = False
        for cx in self.ctxs:
            cx.set_level_fractions(None)

    def finish(self) -> torch.Tensor:
        """The one synchronisation of a sequence: drains every stream, raises on sticky device errors, returns the
        used part of the metric table (device tensor [rows, 8])."""
        with torch.cuda.device(self.device):
            for st in self.streams:
                if st is not self.main:
                    self.main.wait_stream(st)
            first = None
            for cx, st in zip(self.ctxs, self.streams):
                try:
                    cx.check_errors(st.cuda_stream)        # synchronises the stream
                except Exception as e:                     # keep draining: every context must see its own flag
                    first = first or e
            if first is not None:
                raise first
        return self.table[: self.rows_used] if self.table is not None else None

    # ---- whole sequences ---------------------------------------------------------------------------------------------
    def run_sequence(self, batches, n_batches: int = 1) -> np.ndarray:
        """All batches of a sequence through the pipeline; returns the per-scan metric sums [len * n_batches, 8]."""
        from ._native import ERR_NOMEM, SpsError
        batches = list(batches)
        for attempt in (0, 1):
            self.reset_table(len(batches) * n_batches)
            for b in batches:
                self.submit(b, n_batches)
            try:
                return self.finish().cpu().numpy()
            except SpsError as e:
                if e.code != ERR_NOMEM or attempt:
                    raise
                # some cloud outgrew the compact arenas: this is not LiDAR-like data -- every context goes back to
                # full-size arenas (which cannot overflow) and the sequence runs again
                self.use_full_arenas()


def per_scan_metrics(sums: np.ndarray) -> list[dict]:
    """rows of metric sums -> the six per-scan numbers predict_step appends (models.py:88-105)."""
    return [metrics_from_sums(r) for r in np.asarray(sums, dtype=np.float64).reshape(-1, 8)]
