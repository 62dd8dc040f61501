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


def _check_world(tmp_path, world, n_scans):
    sys.path.insert(0, ROOT)
    from sps_amd import parallel
    out = str(tmp_path / f"r0_w{world}_{n_scans}.pt")
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, n_scans, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)
    want_rows = _rows_for(range(n_scans))
    np.testing.assert_array_equal(got["rows"].numpy(), want_rows.numpy())
    want = parallel.mean_metrics(parallel.gather_metric_rows(want_rows, 1))
    for k, v in want.items():
        assert got["mean"][k] == v or (np.isnan(v) and np.isnan(got["mean"][k])), k


def test_shard_and_gather_world8_uneven(tmp_path):
    """BASELINE config 5's world size: 8 ranks, 21 scans (ranks hold 3,3,3,3,3,2,2,2 rows: the padded all-gather drops
    three pad rows) -- rank 0's table and means equal the one-process result."""
    sys.path.insert(0, ROOT)
    from sps_amd import parallel
    assert [len(parallel.shard_indices(21, r, 8)) for r in range(8)] == [3, 3, 3, 3, 3, 2, 2, 2]
    _check_world(tmp_path, 8, 21)


def test_shard_and_gather_world8_ranks_without_scans(tmp_path):
    """Fewer scans than ranks (5 scans on 8 ranks): three ranks contribute only pad rows."""
    _check_world(tmp_path, 8, 5)


RANK_SCRIPT = '''
import json, os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from sps_amd import parallel
n_scans = int(sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = parallel.shard_indices(n_scans, rank, world)
rows = torch.zeros((len(mine), parallel.ROW), dtype=torch.float64)
rows[:, 0] = torch.tensor(mine, dtype=torch.float64)
rows[:, 1] = 10.0 + rows[:, 0]
out = parallel.gather_metric_rows(rows, world)
if rank == 0:
    print(json.dumps({{"world": world, "idx": out[:, 0].tolist(), "count": out[:, 1].tolist()}}))
dist.barrier()
dist.destroy_process_group()
'''


def test_spawn_ranks_starts_the_drivers_command_line_by_itself(tmp_path):
    """parallel.spawn_ranks -- what `python bench.py --gpus N` and `scripts/predict.py --gpus N` call when no launcher set
    WORLD_SIZE: N ranks under torch.distributed.run in a child process, rendezvous on 127.0.0.1, exit code relayed.  Eight
    ranks (config 5's world size), 13 scans: rank 0 prints the complete table once."""
    import json
    import subprocess
    script = tmp_path / "rank_probe.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT))
    driver = ("import sys; sys.path.insert(0, %r); from sps_amd import parallel; "
              "raise SystemExit(parallel.spawn_ranks(8, %r, ['13']))" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["world"] == 8 and d["idx"] == [float(i) for i in range(13)] and d["count"] == [10.0 + i for i in range(13)]
    bad = subprocess.run([sys.executable, "-c", driver.replace("['13']", "['not-a-number']")], capture_output=True, text=True,
                         timeout=300, env=env)
    assert bad.returncode != 0                                  # a failing rank fails the caller
