#!/usr/bin/env python3
"""Training on the MI355X path -- the reference's ``scripts/train.py`` (same ``--config/-c`` option, same data module,
model, optimiser and scheduler, scripts/train.py:30-66) with the Lightning ``Trainer.fit`` loop written out: per batch
``SPSNet.training_step`` (train-mode forward + native backward through libsps_hip.so) -> Adam step; per epoch
``validation_step`` over the validation split, StepLR step, and a Lightning-layout checkpoint (``{"state_dict": ...,
"hyper_parameters": cfg}`` with the ``model.MinkUNet.*`` keys ``scripts/predict.py -w`` and ``util.load_model`` read):
``last.ckpt`` every epoch and the best-``val_loss`` one (ModelCheckpoint(monitor="val_loss", save_last=True),
train.py:37-42).

Under ``python -m torch.distributed.run --nproc-per-node W`` the batches are sharded i mod W over the GPUs and the ranks
average ONE flat gradient tensor per step (a single RCCL all-reduce of 7.4 MB); the reference trains on one GPU.

``--synthetic N`` trains on N synthetic scans (no $DATA tree exists in this environment): labels are a smooth function
of position so that there is something to learn.
"""
from __future__ import annotations

import os
import sys
import time

import click
import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sps.datasets.blt_dataset as datasets  # noqa: E402
import sps.models.models as models  # noqa: E402
from sps_amd import scalars, synthetic  # noqa: E402

DEFAULT_CONFIG_PATH = "./config/config.yaml"
LOG_DIR = "./tb_logs"


def synthetic_loaders(n, voxel_size):
    """N training + N // 4 validation batches [n_i, 6]; label = stability-like score in [0, 1] that depends on height."""
    def relabel(b):
        b = b.copy()
        scan = b[:, 4] == 1
        b[scan, 5] = np.clip(0.5 + 0.25 * b[scan, 3], 0.0, 1.0)          # ground (z ~ -1.8) -> 0.05, walls rise to 1
        return torch.from_numpy(b)
    seq = [relabel(b) for b in synthetic.make_sequence(n + max(1, n // 4), voxel_size=voxel_size, n_azimuth=600, n_beams=32)]
    return seq[:n], seq[n:]


@click.command()
@click.option("--config", "-c", type=str, help="Path to the config file (.yaml)", default=DEFAULT_CONFIG_PATH)
@click.option("--synthetic", "n_synth", type=int, default=0, help="train on N synthetic scans instead of $DATA")
@click.option("--max-epochs", type=int, default=None, help="override TRAIN.MAX_EPOCH")
@click.option("--out", type=str, default=LOG_DIR, help="directory for checkpoints")
@click.option("--host-items", is_flag=True, help="assemble the items on the host (DataLoader workers + scipy cKDTree, as the "
                                                 "reference does) instead of on the device")
def main(config, n_synth, max_epochs, out, host_items):
    cfg = yaml.safe_load(open(config))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("SPS_DIST_BACKEND", "nccl")          # gloo only to exercise the control flow on one GPU
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from sps_amd import hostplace
    hostplace.bind_to_gpu_numa(local)          # CPUs (and pinned buffers) of the GPU's own NUMA node
    if world > 1:
        # data-parallel over the GPUs of one node: batch i -> rank i mod W, identical initial weights (same seed), the
        # flat gradient averaged by ONE all-reduce per step (RCCL over xGMI; sps_amd/models/models.py::_TrainForward)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.manual_seed(0)

    # Load data and model (train.py:33-35)
    train_sampler = None          # DistributedSampler of the host path under torchrun
    loaders_sharded = False       # every rank iterates only its own batches; else batch i -> rank i mod W of a common list
    if n_synth:
        train_loader, val_loader = synthetic_loaders(n_synth, cfg["MODEL"]["VOXEL_SIZE"])
    else:
        data = datasets.BacchusModule(cfg)
        if not host_items:
            # the per-item work (radius submap, stacking, collate, augmentation) runs on the GPU: no KD-trees, no workers;
            # under torchrun every rank assembles only its own batches
            data.train_loader = datasets.DeviceItemLoader(cfg, data.train_scans, data.map, split="train", shuffle=cfg["DATA"]["SHUFFLE"],
                                                          device=dev, shard=(rank, world), even_shards=True)
            data.valid_loader = datasets.DeviceItemLoader(cfg, data.val_scans, data.map, shuffle=False, device=dev,
                                                          shard=(rank, world))
            loaders_sharded = world > 1
        else:
            data.setup()
        train_loader, val_loader = data.train_dataloader(), data.val_dataloader()
        if world > 1 and host_items:
            loaders_sharded = True
            # every rank LOADS only its own shard (the per-item KD-tree query and the augmentation are the expensive part of
            # an item): a DistributedSampler over the training set (same number of samples on every rank, reshuffled per
            # epoch from the epoch number), every W-th item of the validation set
            from torch.utils.data import DataLoader, Subset
            from torch.utils.data.distributed import DistributedSampler
            tds, vds = train_loader.dataset, val_loader.dataset
            train_sampler = DistributedSampler(tds, num_replicas=world, rank=rank, shuffle=bool(cfg["DATA"]["SHUFFLE"]),
                                               seed=0, drop_last=True)
            kw = dict(batch_size=cfg["TRAIN"]["BATCH_SIZE"], collate_fn=data.collate_fn, num_workers=cfg["DATA"]["NUM_WORKER"],
                      pin_memory=True, drop_last=False, timeout=0)
            train_loader = DataLoader(tds, sampler=train_sampler, **kw)
            val_loader = DataLoader(Subset(vds, list(range(rank, len(vds), world))), shuffle=False, **kw)
    model = models.SPSNet(cfg).to(dev)
    (optimizer,), (scheduler,) = model.configure_optimizers()
    epochs = max_epochs if max_epochs is not None else cfg["TRAIN"]["MAX_EPOCH"]
    ckpt_dir = os.path.join(out, cfg["EXPERIMENT"]["ID"], "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)

    def save(path, epoch, val_loss):
        if rank != 0:
            return
        torch.save({"epoch": epoch, "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "hyper_parameters": cfg, "val_loss": val_loss}, path)

    # the scalars the reference logs through Lightning (models.py:74-75,80-81; LearningRateMonitor, TensorBoardLogger:
    # train.py:38,46-50) -> <out>/<ID>/version_<n>/metrics.csv; device tensors are only read at the end of an epoch
    slog = scalars.ScalarLog(out, cfg["EXPERIMENT"]["ID"]) if rank == 0 else None
    global_step = 0
    best = float("inf")
    for epoch in range(epochs):
        model.train()
        t0, losses = time.time(), []
        sharded = loaders_sharded
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        n_even = len(train_loader) if sharded else len(train_loader) // world * world     # the same number of steps on every rank
        for i, batch in enumerate(train_loader):
            if i >= n_even:
                break
            if not sharded and i % world != rank:
                continue
            optimizer.zero_grad(set_to_none=True)
            out_ = model.training_step(batch.to(dev, non_blocking=True), i)
            out_["loss"].backward()
            optimizer.step()
            losses.append(out_["loss"].detach())
            if slog is not None:
                slog.log(epoch, global_step, train_loss=losses[-1], train_r2=out_["val_r2"].detach(),
                         **{"lr-Adam": optimizer.param_groups[0]["lr"]})
            global_step += 1
        train_loss = float(torch.stack(losses).mean()) if losses else float("nan")
        model.eval()
        if world > 1:
            # the parameters are identical on every rank (same initial weights, averaged gradients); the BatchNorm running
            # statistics are not (every rank saw its own batches): validation and the checkpoint use rank 0's
            import torch.distributed as dist
            for buf in model.buffers():
                dist.broadcast(buf, 0)
            model.model.mark_weights_dirty()
        vl, vr = [], []
        with torch.no_grad():
            for i, batch in enumerate(val_loader):
                v = model.validation_step(batch.to(dev, non_blocking=True), i)
                vl.append(v["val_loss"])
                vr.append(v["val_r2"])
                if slog is not None:
                    slog.log(epoch, global_step, val_loss=v["val_loss"], val_r2=v["val_r2"])
        scheduler.step()
        # sticky device errors of the epoch's forwards (coordinate range, a level with a single row under train-mode BatchNorm)
        # surface here: one synchronisation per epoch, not per step
        models.get_context(dev.index or 0, torch.cuda.current_stream().cuda_stream).check_errors(torch.cuda.current_stream().cuda_stream)
        if world > 1 and sharded:                       # every rank validated its own part of the split: pool the sums
            t = torch.tensor([float(torch.stack(vl).sum()) if vl else 0.0, float(torch.stack(vr).sum()) if vr else 0.0,
                              float(len(vl))], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            val_loss = float(t[0] / t[2]) if float(t[2]) > 0 else float("nan")
            val_r2 = float(t[1] / t[2]) if float(t[2]) > 0 else float("nan")
        else:                                           # (the synthetic lists are validated whole on every rank)
            val_loss = float(torch.stack(vl).mean()) if vl else float("nan")
            val_r2 = float(torch.stack(vr).mean()) if vr else float("nan")
        if world > 1:
            t = torch.tensor([train_loss, float(len(losses))], dtype=torch.float64, device=dev)
            t[0] *= t[1]
            dist.all_reduce(t)
            train_loss = float(t[0] / max(float(t[1]), 1.0))
        if rank == 0:
            print(f"epoch {epoch:03d}  train_loss {train_loss:.5f}  val_loss {val_loss:.5f}  val_r2 {val_r2:.4f}  "
                  f"lr {optimizer.param_groups[0]['lr']:.2e}  {len(losses) * world / max(time.time() - t0, 1e-9):.1f} steps/s")
        if slog is not None:
            slog.flush()
        save(os.path.join(ckpt_dir, "last.ckpt"), epoch, val_loss)
        if val_loss < best:
            best = val_loss
            save(os.path.join(ckpt_dir, f"{cfg['EXPERIMENT']['ID']}_epoch={epoch:03d}_val_loss={val_loss:.4f}.ckpt"), epoch, val_loss)
    if rank == 0:
        print("best val_loss", best, "checkpoints in", ckpt_dir)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
