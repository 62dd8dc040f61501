"""The N > 1 path on CPU: two gloo processes shard the scans i mod W, compute metric rows, all-gather
them once; rank 0's mean equals the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows_for(indices):
    rows = []
    for i in indices:
        rng = np.random.default_rng(1000 + i)
        n = 50 + i
        tp, fp, fn = rng.integers(0, 10, 3)
        tn = n - tp - fp - fn
        g = rng.uniform(0, 1, n)
        s = rng.uniform(0, 1, n)
        rows.append([float(i), n, tp, fp, fn, tn, float(np.sum((s - g) ** 2)), float(g.sum()), float((g * g).sum())])
    return torch.tensor(rows, dtype=torch.float64).reshape(-1, 9)


def _worker(rank, world, port, n_scans, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sps_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = parallel.shard_indices(n_scans, rank, world)
    rows = parallel.gather_metric_rows(_rows_for(mine), world)
    if rank == 0:
        torch.save({"rows": rows, "mean": parallel.mean_metrics(rows)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    sys.path.insert(0, ROOT)
    from sps_amd import parallel
    n_scans = 7                                         # odd: ranks hold 4 and 3 rows (padded gather)
    assert parallel.shard_indices(n_scans, 0, 2) == [0, 2, 4, 6] and parallel.shard_indices(n_scans, 1, 2) == [1, 3, 5]
    out = str(tmp_path / "r0.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n_scans, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    want_rows = _rows_for(range(n_scans))
    np.testing.assert_array_equal(got["rows"].numpy(), want_rows.numpy())
    want = parallel.mean_metrics(parallel.gather_metric_rows(want_rows, 1))
    for k, v in want.items():
        assert got["mean"][k] == v or (np.isnan(v) and np.isnan(got["mean"][k])), k
