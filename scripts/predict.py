#!/usr/bin/env python3
"""Offline evaluation on the MI355X path -- the reference's ``scripts/predict.py`` (same options
-w/--weights, -seq/--sequence, -c/--config; same six printed lines), with the Lightning Trainer loop
(predict.py:64-67) replaced by a plain per-scan loop over SPSNet.predict_step.

Differences, on purpose:
  * ``--sequence`` is taken as ONE sequence id (the reference wraps the string in list(), which splits
    it into characters and trips its own assert, predict.py:44-48);
  * ``--synthetic N`` evaluates N synthetic scans (no $DATA tree / checkpoint exist in this environment);
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

DEFAULT_CONFIG_PATH = "./config/config.yaml"


def synthetic_scans(n, voxel_size):
    map_points = synthetic.build_map()
    for i in range(n):
        sc = synthetic.make_scene(scan_seed=100 + i, x_offset=0.5 * i - 2.0, voxel_size=voxel_size,
                                  map_points=map_points)
        yield torch.from_numpy(sc["batch"])


@click.command()
@click.option("--weights", "-w", type=str, default=None, help="path to checkpoint file (.ckpt) to do inference.")
@click.option("--sequence", "-seq", type=str, default=None,
              help="Run inference on a specific sequence. Otherwise, test split from config is used.")
@click.option("--config", "-c", type=str, default=DEFAULT_CONFIG_PATH, help="Path to the config file (.yaml)")
@click.option("--synthetic", "n_synth", type=int, default=0, help="evaluate N synthetic scans instead of $DATA")
def main(weights, sequence, config, n_synth):
    cfg = yaml.safe_load(open(config))
    if sequence:
        cfg["DATA"]["SPLIT"]["TEST"] = [sequence]
    print('Test seq: ', cfg["DATA"]["SPLIT"]["TEST"])
    assert len(cfg["DATA"]["SPLIT"]["TEST"]) == 1, "Only one test SEQ is allowed at a time!"
    cfg["TRAIN"]["BATCH_SIZE"] = 1

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if n_synth:
        scans = list(synthetic_scans(n_synth, cfg["MODEL"]["VOXEL_SIZE"]))
        loader = scans
    else:
        data = datasets.BacchusModule(cfg, test=True)
        data.setup()
        loader = data.test_dataloader()

    model = models.SPSNet(cfg, len(loader))
    if weights:
        ckpt = torch.load(weights, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["state_dict"])
    else:
        print("no --weights given: random-init weights (resnet.py:87-94 scheme)")
    model = model.to(dev).eval().freeze()

    # per-scan rows [scan_idx, Loss, R2, dIoU, Precision, Recall, F1, 0, 0] of this rank's shard
    rows = []
    with torch.no_grad():
        for i, batch in enumerate(loader):
            if i % world != rank:                      # scan i -> rank i mod W (parallel.shard_indices)
                continue
            m = model.predict_step(batch.to(dev, non_blocking=True), i)
            rows.append([float(i), m["loss"], m["r2"], m["dIoU"], m["precision"], m["recall"], m["f1"], 0.0, 0.0])
    local_rows = torch.tensor(rows, dtype=torch.float64, device=dev).reshape(-1, parallel.ROW)
    gathered = parallel.gather_metric_rows(local_rows, world)      # one RCCL all-gather per sequence
    if rank == 0:
        print('\n########## Inference Metrics ##########')
        for j, name in enumerate(["Loss", "R2", "dIoU", "Precision", "Recall", "F1"]):
            col = gathered[:, 1 + j].cpu().numpy()
            mean_value = float(np.sum(col) / max(len(col), 1))     # mean of per-scan values (predict.py:80-83)
            print(f'{name} {"." * (12 - len(name))} {mean_value:.3f}')
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
