#!/usr/bin/env python3
"""Training on the MI355X path -- the reference's ``scripts/train.py`` (same ``--config/-c`` option, same data module,
model, optimiser and scheduler, scripts/train.py:30-66) with the Lightning ``Trainer.fit`` loop written out: per batch
``SPSNet.training_step`` (train-mode forward + native backward through libsps_hip.so) -> Adam step; per epoch
``validation_step`` over the validation split, StepLR step, and a Lightning-layout checkpoint (``{"state_dict": ...,
"hyper_parameters": cfg}`` with the ``model.MinkUNet.*`` keys ``scripts/predict.py -w`` and ``util.load_model`` read):
``last.ckpt`` every epoch and the best-``val_loss`` one (ModelCheckpoint(monitor="val_loss", save_last=True),
train.py:37-42).

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
from sps_amd import synthetic  # noqa: E402

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
def main(config, n_synth, max_epochs, out):
    cfg = yaml.safe_load(open(config))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)

    # Load data and model (train.py:33-35)
    if n_synth:
        train_loader, val_loader = synthetic_loaders(n_synth, cfg["MODEL"]["VOXEL_SIZE"])
    else:
        data = datasets.BacchusModule(cfg)
        data.setup()
        train_loader, val_loader = data.train_dataloader(), data.val_dataloader()
    model = models.SPSNet(cfg).to(dev)
    (optimizer,), (scheduler,) = model.configure_optimizers()
    epochs = max_epochs if max_epochs is not None else cfg["TRAIN"]["MAX_EPOCH"]
    ckpt_dir = os.path.join(out, cfg["EXPERIMENT"]["ID"], "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)

    def save(path, epoch, val_loss):
        torch.save({"epoch": epoch, "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "hyper_parameters": cfg, "val_loss": val_loss}, path)

    best = float("inf")
    for epoch in range(epochs):
        model.train()
        t0, losses = time.time(), []
        for i, batch in enumerate(train_loader):
            optimizer.zero_grad(set_to_none=True)
            out_ = model.training_step(batch.to(dev, non_blocking=True), i)
            out_["loss"].backward()
            optimizer.step()
            losses.append(out_["loss"].detach())
        train_loss = float(torch.stack(losses).mean()) if losses else float("nan")
        model.eval()
        vl, vr = [], []
        with torch.no_grad():
            for i, batch in enumerate(val_loader):
                v = model.validation_step(batch.to(dev, non_blocking=True), i)
                vl.append(v["val_loss"])
                vr.append(v["val_r2"])
        val_loss = float(torch.stack(vl).mean()) if vl else float("nan")
        val_r2 = float(torch.stack(vr).mean()) if vr else float("nan")
        scheduler.step()
        print(f"epoch {epoch:03d}  train_loss {train_loss:.5f}  val_loss {val_loss:.5f}  val_r2 {val_r2:.4f}  "
              f"lr {optimizer.param_groups[0]['lr']:.2e}  {len(losses) / max(time.time() - t0, 1e-9):.1f} steps/s")
        save(os.path.join(ckpt_dir, "last.ckpt"), epoch, val_loss)
        if val_loss < best:
            best = val_loss
            save(os.path.join(ckpt_dir, f"{cfg['EXPERIMENT']['ID']}_epoch={epoch:03d}_val_loss={val_loss:.4f}.ckpt"), epoch, val_loss)
    print("best val_loss", best, "checkpoints in", ckpt_dir)


if __name__ == "__main__":
    main()
