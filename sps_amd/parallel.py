"""Data-parallel sharding of scans over the GPUs of one node.

The reference is single process / single GPU (scripts/predict.py:64 ``devices=1``); every scan (+ its
submap) is an independent forward, so the path shards with NO data-path collective: one process per
GPU, scan i -> rank i mod W, weights and map hash replicated.  The only exchange is one all-gather
of the per-scan metric rows at the end of a sequence (RCCL over xGMI: ``torch.distributed`` backend
"nccl" on ROCm; "gloo" in the CPU tests).  Rows are [scan_idx, count, TP, FP, FN, TN, sse, sum g,
sum g^2] float64; rank 0 sorts by scan index and takes the MEAN OF PER-SCAN metrics exactly as
predict.py:80-83 does (not pooled counts).
"""
from __future__ import annotations

import torch

ROW = 9  # scan_idx + the 8 sums of sps_metrics


def shard_indices(n_scans: int, rank: int, world: int):
    """Scan indices owned by `rank`: i mod W == rank (keeps the per-rank load even within a sequence)."""
    return list(range(rank, n_scans, world))


def spawn_ranks(n: int, script: str, argv) -> int:
    """Run ``script argv`` as n ranks of one node under ``torch.distributed.run`` -- the driver's N > 1 command line -- in a
    CHILD process and return its exit code.  For entry points called as ``python bench.py --gpus N`` without a launcher:
    the caller must not have touched the GPU yet (a child, never an exec of a process that holds HIP state).  The
    rendezvous is 127.0.0.1 on a free port (the container's hostname may not resolve)."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    return subprocess.call(cmd, env=env)


def gather_metric_rows(local_rows: torch.Tensor, world: int, group=None, force: bool = False) -> torch.Tensor:
    """All-gather the ranks' [n_local, ROW] rows (n_local may differ by one) -> [n_total, ROW] sorted by
    scan index.  One padded all_gather; the pad rows carry scan_idx = -1 and are dropped.  ``force`` runs the
    collective at world size 1 too (an initialised one-rank group: exercises the RCCL path on a single-GPU box)."""
    import torch.distributed as dist
    assert local_rows.dim() == 2 and local_rows.shape[1] == ROW
    if world == 1 and not force:
        out = local_rows
    else:
        n_local = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
        counts = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(counts, n_local, group=group)
        n_max = int(max(int(c.item()) for c in counts))
        padded = torch.full((n_max, ROW), -1.0, dtype=local_rows.dtype, device=local_rows.device)
        padded[: local_rows.shape[0]] = local_rows
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        out = torch.cat(parts, 0)
        out = out[out[:, 0] >= 0]
    order = torch.argsort(out[:, 0])
    return out[order]


def mean_metrics(rows: torch.Tensor) -> dict:
    """Mean over scans of the per-scan metrics (predict.py:70-83): Loss, R2, dIoU, Precision, Recall, F1."""
    from .models.models import metrics_from_sums
    per = [metrics_from_sums(r[1:].tolist()) for r in rows.cpu()]
    names = {"Loss": "loss", "R2": "r2", "dIoU": "dIoU", "Precision": "precision", "Recall": "recall", "F1": "f1"}
    n = max(len(per), 1)
    return {k: sum(m[v] for m in per) / n for k, v in names.items()}
