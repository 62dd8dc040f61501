#!/usr/bin/env python3
"""Offline evaluation on the MI355X path -- the reference's ``scripts/predict.py`` (same options
-w/--weights, -seq/--sequence, -c/--config; same six printed lines), with the Lightning Trainer loop
(predict.py:64-67) replaced by the stream-pipelined sps_amd.engine.ScanEngine (the loop bench.py measures): per
scan one host->device copy + one fused forward+metrics call, ONE host synchronisation per sequence.

Differences, on purpose:
  * ``--sequence`` is taken as ONE sequence id (the reference wraps the string in list(), which splits
    it into characters and trips its own assert, predict.py:44-48);
  * ``--synthetic N`` evaluates N synthetic scans (no $DATA tree / checkpoint exist in this environment);
  * the per-scan item (scan rows + KD-tree radius submap, blt_dataset.py:209-271) is assembled ON THE DEVICE from the
    cached scans (sps_radius_item; ``--host-items`` keeps the reference's DataLoader + scipy cKDTree path);
  * launched under ``python -m torch.distributed.run --nproc-per-node W`` the scans are sharded
    i mod W over the GPUs and the per-scan metric rows are all-gathered once (RCCL).
"""
from __future__ import annotations

import os
import sys

import click
import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sps.datasets.blt_dataset as datasets  # noqa: E402
import sps.models.models as models  # noqa: E402
from sps_amd import parallel, synthetic  # noqa: E402
from sps_amd.engine import ScanEngine  # noqa: E402

DEFAULT_CONFIG_PATH = "./config/config.yaml"


def synthetic_scans(n, voxel_size):
    """BASELINE config 3's sequence: consecutive scans, the sensor advancing 0.5 m per scan; pinned host tensors, as the
    reference's DataLoader (pin_memory=True, blt_dataset.py:102-118) would deliver them."""
    for b in synthetic.make_sequence(n, voxel_size=voxel_size):
        t = torch.from_numpy(b)
        yield t.pin_memory() if torch.cuda.is_available() else t


def batched(loader, k):
    """k consecutive single-scan batches -> one [sum N, 6] tensor with batch column 0..k-1 (collate_fn layout)."""
    group = []
    for b in loader:
        group.append(b)
        if len(group) == k:
            yield group
            group = []
    if group:
        yield group


@click.command()
@click.option("--weights", "-w", type=str, default=None, help="path to checkpoint file (.ckpt) to do inference.")
@click.option("--sequence", "-seq", type=str, default=None,
              help="Run inference on a specific sequence. Otherwise, test split from config is used.")
@click.option("--config", "-c", type=str, default=DEFAULT_CONFIG_PATH, help="Path to the config file (.yaml)")
@click.option("--synthetic", "n_synth", type=int, default=0, help="evaluate N synthetic scans instead of $DATA")
@click.option("--batch-size", "-b", "batch_size", type=int, default=1,
              help="scans per forward (batch column 0..b-1, BacchusModule.collate_fn layout).  Default 1 = the reference "
                   "(it forces BATCH_SIZE = 1, predict.py:50).  The metric sums are kept per batch index, so any value prints "
                   "the same per-scan means up to the f32 summation order inside a forward (scores move by <= 2e-6: a label "
                   "exactly at the threshold can flip); -b 4 (BASELINE config 3) is the opt-in throughput mode: ~15 %% more "
                   "scans/s")
@click.option("--me-conventions", "me_conventions", type=str, default=None,
              help="MinkowskiEngine conventions the checkpoint follows, 'option=value,...' (sps_amd/conventions.py; "
                   "tools/convention_probe.py finds them); default: MODEL.ME_CONVENTIONS of the config, else the canonical ones")
@click.option("--streams", type=int, default=None, help="forwards in flight (HIP streams); default: the engine's")
@click.option("--timing", is_flag=True, help="print scans/s of the evaluation loop (rank 0)")
@click.option("--force-dist", is_flag=True, help="initialise the process group (RCCL) and all-gather the metric rows even at world size 1")
@click.option("--backend", type=str, default=None,
              help="torch.distributed backend (default: $SPS_DIST_BACKEND, else nccl = RCCL over xGMI).  gloo lets several ranks "
                   "share ONE GPU (rank -> device LOCAL_RANK mod device count): how the sharding and the padded metric "
                   "all-gather are tested on a one-GPU box")
@click.option("--gpus", type=int, default=None,
              help="data-parallel ranks, one per GPU (BASELINE config 5).  Without a launcher in the environment the ranks are "
                   "started here (python -m torch.distributed.run --nproc-per-node N, a child process); under torchrun it "
                   "must equal WORLD_SIZE")
@click.option("--host-items", is_flag=True, help="assemble the items on the host (DataLoader workers + scipy cKDTree, as the "
                                                 "reference does) instead of on the device")
def main(weights, sequence, config, n_synth, batch_size, me_conventions, streams, timing, force_dist, backend, gpus, host_items):
    if gpus and gpus > 1 and "WORLD_SIZE" not in os.environ:        # nothing has touched the GPU yet: start the ranks
        raise SystemExit(parallel.spawn_ranks(gpus, os.path.abspath(__file__), sys.argv[1:]))
    assert not gpus or gpus == int(os.environ.get("WORLD_SIZE", "1")), "--gpus does not match WORLD_SIZE"
    cfg = yaml.safe_load(open(config))
    if me_conventions:
        cfg["MODEL"]["ME_CONVENTIONS"] = me_conventions
    if sequence:
        cfg["DATA"]["SPLIT"]["TEST"] = [sequence]
    print('Test seq: ', cfg["DATA"]["SPLIT"]["TEST"])
    assert len(cfg["DATA"]["SPLIT"]["TEST"]) == 1, "Only one test SEQ is allowed at a time!"
    cfg["TRAIN"]["BATCH_SIZE"] = 1
    assert 1 <= batch_size <= 31, "batch size must be in [1, 31]"

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = backend or os.environ.get("SPS_DIST_BACKEND", "nccl")
    if backend != "nccl":                          # ranks may share a GPU (RCCL wants one device per rank)
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from sps_amd import hostplace
    hostplace.bind_to_gpu_numa(local)          # CPUs (and pinned buffers) of the GPU's own NUMA node
    use_dist = world > 1 or force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    raw_scans = None
    if n_synth:
        loader = list(synthetic_scans(n_synth, cfg["MODEL"]["VOXEL_SIZE"]))
    else:
        data = datasets.BacchusModule(cfg, test=True)           # $DATA tree -> cached scans in the map frame + the map
        if host_items:
            data.setup()                                        # KD-tree of the map, DataLoader (blt_dataset.py:102-118,198)
            loader = data.test_dataloader()
        else:
            raw_scans = loader = data.test_scans                # the items are assembled on the device (no KD-trees at all)
    n_scans = len(loader)

    if not weights:
        torch.manual_seed(0)                       # reproducible random init (the same on every rank)
    model = models.SPSNet(cfg, n_scans)
    if weights:
        ckpt = torch.load(weights, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["state_dict"])
    else:
        print("no --weights given: random-init weights (resnet.py:87-94 scheme, torch.manual_seed(0))")
    model = model.to(dev).eval().freeze()

    # The reference's loop (predict.py:64-67: trainer.predict -> predict_step per scan, three host syncs each) as a
    # stream-ordered pipeline: group g of `batch_size` scans -> rank g mod W; every group is copied to the device and
    # evaluated on one of the engine's streams; the 8 metric sums of every scan land in a device table; ONE
    # synchronisation at the end of the sequence.
    kw = {} if streams is None else {"streams": streams}
    # arenas and staging buffers of every stream are sized before the loop (largest group of the sequence when it is in
    # memory, else batch_size x the first scan with 50 % head room; a larger cloud later only costs one re-allocation)
    if raw_scans is not None:
        groups = None
        max_rows = 0
    elif isinstance(loader, list):
        # the synthetic sequence is in memory: collate the groups now, into pinned tensors -- what the reference's DataLoader
        # (batch_size, collate_fn, pin_memory=True; worker processes) hands to the loop
        groups = []
        for g in batched(loader, batch_size):
            b = g[0] if len(g) == 1 else datasets.BacchusModule.collate_fn([x[:, 1:] for x in g])
            groups.append((b.pin_memory() if (dev.type == "cuda" and not b.is_pinned()) else b, len(g)))
        max_rows = max(int(b.shape[0]) for b, _ in groups) if groups else 0
    else:
        groups = None
        first = next(iter(loader))
        max_rows = int(first.shape[0] * batch_size * 1.5)
    eng = ScanEngine(model, dev, table_rows=n_scans + batch_size, max_rows=max_rows, stage_cols=6, **kw)
    if raw_scans is not None:
        # the map goes to the device once (uniform cell grid of the radius query); buffers and arenas of every stream are
        # sized for the largest group of the sequence before the loop
        eng.attach_map(data.map[:, :3], cfg["MODEL"]["VOXEL_SIZE"])
        eng.calibrate_rows(raw_scans[:: max(1, len(raw_scans) // 4)][:4])          # item rows per scan point of this map
        eng.prepare_scans(max(sum(len(s) for s in g) for g in batched(raw_scans, batch_size)), raw_scans[0].dtype)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from sps_amd._native import ERR_ITEMCAP, ERR_NOMEM, SpsError
    for attempt in range(5):
        idx = []                                          # scan index of every used table row
        eng.reset_table(n_scans + batch_size)
        with torch.no_grad():
            if raw_scans is not None:
                for g, grp in enumerate(batched(raw_scans, batch_size)):
                    if g % world != rank:
                        continue
                    eng.submit_scans(grp)
                    idx += [g * batch_size + j for j in range(len(grp))]
            it = () if raw_scans is not None else groups if groups is not None else (
                (grp[0] if len(grp) == 1 else datasets.BacchusModule.collate_fn([b[:, 1:] for b in grp]), len(grp))
                for grp in batched(loader, batch_size))
            for g, (batch, ng) in enumerate(it):
                if g % world != rank:                     # parallel.shard_indices over groups
                    continue
                eng.submit(batch, ng)
                idx += [g * batch_size + j for j in range(ng)]
        try:
            sums = eng.finish()
            break
        except SpsError as e:
            # a cloud whose coarse levels do not thin out like a LiDAR scan's outgrew the compact arenas (its forward was
            # aborted): evaluate the sequence again on full-size arenas
            if e.code not in (ERR_NOMEM, ERR_ITEMCAP) or attempt == 4:
                raise
            if e.code == ERR_ITEMCAP:
                # the map is denser around some scan than the item buffers were sized for (calibrated on the first group)
                if rank == 0:
                    print("an item outgrew the item buffers: evaluating the sequence again with larger buffers", file=sys.stderr)
                eng.row_factor *= 1.5
            else:
                if rank == 0:
                    print("a cloud outgrew the LiDAR-sized arenas: evaluating the sequence again on full-size arenas", file=sys.stderr)
                eng.use_full_arenas()
    n_local = len(idx)
    local_rows = torch.empty((n_local, parallel.ROW), dtype=torch.float64, device=dev)
    local_rows[:, 0] = torch.tensor(idx, dtype=torch.float64, device=dev)
    local_rows[:, 1:] = sums[:n_local]
    gathered = parallel.gather_metric_rows(local_rows, world, force=use_dist)      # one RCCL all-gather per sequence
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        mean = parallel.mean_metrics(gathered)                     # mean of per-scan values (predict.py:80-83)
        for m in (models.metrics_from_sums(r[1:].tolist()) for r in gathered.cpu()):
            model.predict_loss.append(m["loss"]); model.predict_r2.append(m["r2"]); model.dIoU.append(m["dIoU"])
            model.precision.append(m["precision"]); model.recall.append(m["recall"]); model.F1.append(m["f1"])
        print('\n########## Inference Metrics ##########')
        for name in ["Loss", "R2", "dIoU", "Precision", "Recall", "F1"]:
            print(f'{name} {"." * (12 - len(name))} {mean[name]:.3f}')
        if timing:
            print(f"timing: {len(gathered)} scans in {dt:.3f} s = {len(gathered) / dt:.1f} scans/s "
                  f"({world} GPU(s), {len(eng.streams)} streams, batch {batch_size}, host->device copies included)")
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
