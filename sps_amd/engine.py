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

# Pipelines (HIP streams) per engine.  The runtime deals the streams of a process to 4 hardware queues in creation order, the
# caller's current stream included, and a forward is as fast as its queue is free: the pipelines are spread evenly when their
# number, WITH the current stream as one of them (include_main), is a multiple of 4.  Round 5, one box, 400 steps: 8 pipelines
# incl. the current stream 4 340-4 357 scans/s host-fed (resident 4 367-4 381) against 4 158-4 191 (4 261-4 290) for round 4's
# 7 side streams; 4 incl. the current stream are the best SHORT pipeline (20 steps: 3 887-3 945 against 3 757-3 824), but
# leave a queue idle during a pipeline's host -> device copy (400 steps host-fed: 4 110); 8 side streams WITHOUT the current
# one: 3 929 (profiles/round5_a/streams_queues.txt)
DEFAULT_STREAMS = 8
SHORT_RUN_STREAMS = 4


class ScanEngine:
    def __init__(self, net: SPSNet, device: torch.device | int | None = None, streams: int = DEFAULT_STREAMS,
                 max_rows: int = 0, table_rows: int = 0, stage_cols: int = 0, compact: bool = True, include_main: bool = True,
                 pipelined: bool | None = None):
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
            if S > 1 and include_main:        # the caller's current stream carries one of the S pipelines
                self.streams = [self.main] + [torch.cuda.Stream(device=self.device) for _ in range(S - 1)]
            else:
                self.streams = [torch.cuda.Stream(device=self.device) for _ in range(S)] if S > 1 else [self.main]
            # the current stream's pipeline gets a context of ITS OWN: the shared per-stream context of that stream
            # (get_context) belongs to whoever calls the model directly there and must not become compact / inference-only
            from ._native import Context
            self.ctxs = [Context(self.index) if st is self.main else get_context(self.index, st.cuda_stream)
                         for st in self.streams]
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
            # several forwards in flight: the geometry that does the least work; a single pipeline keeps the shortest chain
            # (`pipelined` overrides: profiling passes run ONE stream with the kernels exactly as the pipelines launch them)
            cx.set_pipelined(S > 1 if pipelined is None else bool(pipelined))
        self._next = 0
        self._stage = [None] * S          # per-stream PAIR of device staging buffers for host batches
        self._pinned = [None] * S         # (pageable host batches only: pinned bounce buffers)
        self._flip = [0] * S
        self._copied = [[None, None] for _ in range(S)]       # event behind the copy into the buffer
        self._done = [None] * S                               # event behind the last host-fed forward of the stream
        # device item path (attach_map / submit_scans): per-stream raw-scan staging (host pinned + device), item rows, row count
        self.submap = None
        self.row_factor = 2.5             # item rows (scan + radius submap, duplicates kept) per scan point the buffers are sized for
        self._raw_pin = [None] * S
        self._raw_dev = [None] * S
        self._rows = [None] * S
        self._nrows = [None] * S
        self._scores = [None] * S
        self._raw_ev = [None] * S
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
                if st is not self.main:
                    st.wait_stream(self.main)          # `warm` / `out` are filled on the main stream
                with torch.cuda.stream(st):
                    self.net.forward_metrics(warm, 1, out, ctx=cx)
            if table_rows:
                self.reset_table(table_rows)
            if stage_cols and max_rows:
                for k in range(len(self.streams)):
                    # allocated ON the stream that uses them: when _to_device drops a pair for a larger one, the caching
                    # allocator returns the blocks to that stream's pool (no other stream can be handed memory a forward
                    # in flight on stream k still reads)
                    with torch.cuda.stream(self.streams[k]):
                        self._stage[k] = [torch.empty((int(max_rows), int(stage_cols)), dtype=torch.float32, device=self.device)
                                          for _ in range(2)]
            torch.cuda.synchronize(self.device)

    def reset_table(self, rows: int) -> None:
        if self.table is None or self.table.shape[0] < rows:
            # every forward writes all 8 sums of its rows (k_points_to_blocks clears them first), so no fill is needed --
            # a zero fill on the main stream would race with the side streams' writes.  The block is allocated on the main
            # stream: the side streams wait for the allocation point and are recorded as users of the block.
            with torch.cuda.device(self.device):
                self.table = torch.empty((int(rows), 8), dtype=torch.float64, device=self.device)
                for st in self.streams:
                    if st is not self.main:
                        st.wait_stream(self.main)
                        self.table.record_stream(st)
        self.rows_used = 0

    # ---- steady state ----------------------------------------------------------------------------------------------
    def submit(self, batch: torch.Tensor, n_batches: int = 1, row: int | None = None) -> torch.Tensor:
        """One step: ``batch`` [N,6] = (b,x,y,z,t,label) rows with b < n_batches (BacchusModule.collate_fn layout).
        A host tensor is copied to the device on the scan's stream (pinned source, asynchronous) into one of the stream's
        TWO staging buffers, with an event recorded right behind the copy.  The metric sums of batch index b go to table row
        ``row + b`` (default: the next free rows).  Returns the scores [N] (device tensor, valid after finish() or a wait
        on stream ``self.last_stream``).

        Measured (tools/streams_h2d_sweep.sh): with the two buffers and the event
        behind the copy the host-fed loop runs within 3 % of the resident-input loop at 7 streams (3 678 vs 3 776 scans/s);
        one staging buffer and no event: 16 % below it.  A dedicated copy stream is worse either way: a device-side wait
        on the copy's event (hipStreamWaitEvent behind an SDMA copy) costs the HOST 0.3 ms per step, a host-side wait
        0.17 ms."""
        k = self._next
        self._next = (k + 1) % len(self.streams)
        st = self.streams[k]
        if row is None:
            row = self.rows_used
        if self.table is None or row + n_batches > self.table.shape[0]:
            raise ValueError("metric table too small: call reset_table(rows) with the number of scans of the sequence")
        self.rows_used = max(self.rows_used, row + n_batches)
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            host_fed = not batch.is_cuda
            if host_fed:
                batch = self._to_device(k, batch)
            scores, _ = self.net.forward_metrics(batch, n_batches, self.table[row: row + n_batches], ctx=self.ctxs[k])
            if host_fed:
                # an event behind the forward as well (measured: without it the host-fed loop runs 14 % slower, 3 170 vs
                # 3 678 scans/s at 7 streams -- the marker makes the runtime hand the stream's pending copy + kernels to
                # the hardware queue at once)
                ev = self._done[k]
                if ev is None:
                    ev = self._done[k] = torch.cuda.Event()
                ev.record(st)
        self.last_stream = st
        return scores

    def _to_device(self, k: int, host: torch.Tensor) -> torch.Tensor:
        """Host batch -> the next of stream k's two device staging buffers, on stream k (the caller's current stream)."""
        n = host.shape[0]
        if host.dtype != torch.float32 or not host.is_contiguous():
            host = host.to(torch.float32).contiguous()
        slot = self._flip[k]
        self._flip[k] = slot ^ 1
        bufs = self._stage[k]
        if bufs is None or bufs[0].shape[0] < n or bufs[0].shape[1] != host.shape[1]:
            cap = max(n, 1024)
            bufs = self._stage[k] = [torch.empty((cap + cap // 4, host.shape[1]), dtype=torch.float32, device=self.device)
                                     for _ in range(2)]
        if not host.is_pinned():
            pin = self._pinned[k]
            if pin is None or pin[0].shape[0] < n or pin[0].shape[1] != host.shape[1]:
                pin = self._pinned[k] = [torch.empty((max(n, bufs[0].shape[0]), host.shape[1]), dtype=torch.float32).pin_memory()
                                         for _ in range(2)]
            if self._copied[k][slot] is not None:
                self._copied[k][slot].synchronize()       # the copy that last read this pinned bounce buffer is done
            pin[slot][:n].copy_(host)
            host = pin[slot][:n]
        dst = bufs[slot][:n]
        dst.copy_(host, non_blocking=True)
        ev = self._copied[k][slot]
        if ev is None:
            ev = self._copied[k][slot] = torch.cuda.Event()
        ev.record(self.streams[k])
        return dst

    # ---- offline items assembled on the device (blt_dataset.py:209-244 + collate_fn :173-182) -----------------------------
    def attach_map(self, map_xyz, radius: float) -> None:
        """The map of the variant-A submap (reference blt_dataset.py:222-226: every map point within ``radius`` of a scan
        point) goes to the device ONCE: a uniform cell grid owned by the first context, attached as a view to the others."""
        from .datasets.blt_dataset import DeviceRadiusSubmap
        with torch.cuda.device(self.device):
            self.submap = DeviceRadiusSubmap(map_xyz, radius, device=self.device, ctx=self.ctxs[0])
            for cx in self.ctxs[1:]:
                cx.radius_grid_attach(self.ctxs[0])

    def calibrate_rows(self, scans) -> float:
        """Sets ``row_factor`` (item rows per scan point the buffers are sized for) from the radius query of a few sample
        scans: 1.25 x the largest (n + m) / n seen (the submap keeps duplicates: a map at the network's voxel size gives
        ~6 rows per scan point).  One-off, synchronises."""
        worst = 1.0
        for s in scans:
            _, counts = self.submap.query(np.asarray(s)[:, :3])
            worst = max(worst, 1.0 + float(counts.sum().item()) / max(len(s), 1))
        self.row_factor = max(self.row_factor, 1.25 * worst)
        return self.row_factor

    def prepare_scans(self, max_points: int, dtype=np.float64) -> None:
        """Raw-scan staging, item buffers and arenas of every stream sized for groups of up to ``max_points`` scan points
        (everything that is not steady state happens before the loop)."""
        f64 = np.dtype(dtype) == np.float64
        tdt = torch.float64 if f64 else torch.float32
        cap_rows = int(max_points * self.row_factor) + 1024
        with torch.cuda.device(self.device):
            for k, (cx, st) in enumerate(zip(self.ctxs, self.streams)):
                with torch.cuda.stream(st):
                    self._raw_pin[k] = torch.empty((max_points + 1024, 4), dtype=tdt).pin_memory()
                    self._raw_dev[k] = torch.empty((max_points + 1024, 4), dtype=tdt, device=self.device)
                    self._raw_ev[k] = torch.cuda.Event()
                    self._rows[k] = torch.empty((cap_rows, 6), dtype=torch.float32, device=self.device)
                    self._scores[k] = torch.empty(cap_rows, dtype=torch.float32, device=self.device)
                    self._nrows[k] = torch.zeros(4, dtype=torch.int32, device=self.device)
                cx.reserve(cap_rows)
            torch.cuda.synchronize(self.device)

    def submit_scans(self, scans, row: int | None = None) -> torch.Tensor:
        """One step from RAW scans: ``scans`` = k arrays [n_j, >= 4] = (x, y, z, label) in the map frame (what
        BacchusModule.cash_scans keeps), one batch index each.  Stream-ordered, no host synchronisation: one pinned
        host -> device copy of the raw rows, per scan ``sps_radius_item`` (scan rows + radius submap rows appended to the
        stream's item buffer, row count on the device), then ``sps_forward_metrics_n``.  Replaces the DataLoader worker's
        per-item KD-tree query (blt_dataset.py:224-226,258-271), add_timestamp / stacking (:209-244) and collate_fn.
        Returns the scores buffer of the stream ([row capacity]; the first ``n_rows`` entries are valid after finish())."""
        if self.submap is None:
            raise RuntimeError("attach_map(map_xyz, radius) first")
        k = self._next
        self._next = (k + 1) % len(self.streams)
        st = self.streams[k]
        nb = len(scans)
        if row is None:
            row = self.rows_used
        if self.table is None or row + nb > self.table.shape[0]:
            raise ValueError("metric table too small: call reset_table(rows) with the number of scans of the sequence")
        self.rows_used = max(self.rows_used, row + nb)
        arrs = [s.numpy() if torch.is_tensor(s) else np.asarray(s) for s in scans]
        f64 = arrs[0].dtype == np.float64          # the radius test runs on the scan's own dtype (cKDTree promotes float32)
        npdt, tdt, esz = (np.float64, torch.float64, 8) if f64 else (np.float32, torch.float32, 4)
        n_tot = int(sum(a.shape[0] for a in arrs))
        cap_rows = int(n_tot * self.row_factor) + 1024
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            pin = self._raw_pin[k]
            if pin is None or pin.dtype != tdt or pin.shape[0] < n_tot:
                st.synchronize()
                grow = n_tot + n_tot // 4 + 1024
                pin = self._raw_pin[k] = torch.empty((grow, 4), dtype=tdt).pin_memory()
                self._raw_dev[k] = torch.empty((grow, 4), dtype=tdt, device=self.device)
                self._raw_ev[k] = torch.cuda.Event()
            elif self._raw_ev[k] is not None:
                self._raw_ev[k].synchronize()     # this stream's previous copy out of the pinned buffer is done
            if self._rows[k] is None or self._rows[k].shape[0] < cap_rows:
                grow = cap_rows + cap_rows // 4
                self._rows[k] = torch.empty((grow, 6), dtype=torch.float32, device=self.device)
                self._scores[k] = torch.empty(grow, dtype=torch.float32, device=self.device)
                if self._nrows[k] is None:
                    self._nrows[k] = torch.zeros(4, dtype=torch.int32, device=self.device)
            host = pin.numpy()
            o = 0
            for a in arrs:
                host[o: o + a.shape[0]] = a[:, :4]          # casts to the group's dtype
                o += a.shape[0]
            raw, rows, nrows, scores = self._raw_dev[k], self._rows[k], self._nrows[k], self._scores[k]
            raw[:n_tot].copy_(pin[:n_tot], non_blocking=True)
            self._raw_ev[k].record(st)
            cx = self.ctxs[k]
            self.net.model._sync_weights(cx)
            o = 0
            for j, a in enumerate(arrs):
                cx.radius_item(raw.data_ptr() + o * 4 * esz, f64, 4, a.shape[0], float(j), None if j == 0 else nrows.data_ptr(),
                               rows.data_ptr(), 6, rows.shape[0], nrows.data_ptr(), st.cuda_stream)
                o += a.shape[0]
            cx.forward_metrics_n(rows.data_ptr(), 6, rows.shape[0], nrows.data_ptr(), self.vs, self.eps, nb, scores.data_ptr(),
                                 self.table[row: row + nb].data_ptr(), st.cuda_stream)
            ev = self._done[k]                      # marker behind the step (see submit())
            if ev is None:
                ev = self._done[k] = torch.cuda.Event()
            ev.record(st)
        self.last_stream = st
        return scores

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
