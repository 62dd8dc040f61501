#!/usr/bin/env python3
"""Capture golden input/output vectors from the REFERENCE python (ibrahimhroob/SPS under
/root/reference) for the parts of the hot path that are importable here:

  sps.datasets.util.calculate_metrics / transform_point_cloud / inverse_transform_point_cloud
  sps.datasets.blt_dataset.BacchusDataset.__getitem__ / select_closest_points, BacchusModule.collate_fn
  sps.datasets.augmentation.* and BacchusDataset.augment_data (training path)

Third-party modules that are absent (rospy, ros_numpy, tf, sensor_msgs, MinkowskiEngine,
pytorch_lightning, torchmetrics) are replaced by empty stubs: nothing that executes them is captured.
Runs ONLY in the build container (the reference never travels); writes tests/golden/*.npz (data only).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("rospy")
    mod("ros_numpy")
    mod("tf")
    mod("tf.transformations", quaternion_matrix=lambda q: None)
    mod("sensor_msgs")
    mod("sensor_msgs.msg", PointCloud2=object, PointField=object)
    me = mod("MinkowskiEngine", SparseTensor=object, TensorField=object, MinkowskiConvolution=object,
             MinkowskiBatchNorm=object)
    mod("MinkowskiEngine.modules")
    mod("MinkowskiEngine.modules.resnet_block", BasicBlock=type("BasicBlock", (), {"expansion": 1}),
        Bottleneck=type("Bottleneck", (), {"expansion": 4}))
    me.modules = sys.modules["MinkowskiEngine.modules"]
    mod("pytorch_lightning", LightningDataModule=object, LightningModule=torch.nn.Module)
    mod("torchmetrics", R2Score=object)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import sps.datasets.util as util
    import sps.datasets.blt_dataset as blt

    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)

    # ---- calculate_metrics (util.py:285-299)
    cases_gt, cases_pred, cases_out = [], [], []
    specs = [(200, 0.5, 0.5), (200, 0.1, 0.9), (50, 0.0, 0.3), (50, 0.4, 0.0), (1, 1.0, 1.0), (64, 1.0, 0.0)]
    with np.errstate(all="ignore"):
        for n, p_gt, p_pred in specs:
            gt = (rng.uniform(size=n) < p_gt).astype(np.int64)
            pred = (rng.uniform(size=n) < p_pred).astype(np.int64)
            res = util.calculate_metrics(gt, pred)
            cases_gt.append(gt)
            cases_pred.append(pred)
            cases_out.append(np.asarray(res, dtype=np.float64))
        gt = np.array([0, 1, 1, 0, 1]); pred = np.array([0, 1, 0, 1, 1])
        cases_gt.append(gt); cases_pred.append(pred)
        cases_out.append(np.asarray(util.calculate_metrics(gt, pred), dtype=np.float64))
    np.savez(os.path.join(OUT, "calculate_metrics.npz"),
             **{f"gt{i}": g for i, g in enumerate(cases_gt)}, **{f"pred{i}": p for i, p in enumerate(cases_pred)},
             **{f"out{i}": o for i, o in enumerate(cases_out)}, n=len(cases_out))

    # ---- transform_point_cloud / inverse (util.py:187-206)
    pts = rng.uniform(-20, 20, size=(300, 3))
    ang = 0.7
    T = np.array([[np.cos(ang), -np.sin(ang), 0, 1.5], [np.sin(ang), np.cos(ang), 0, -2.0], [0, 0, 1, 0.3], [0, 0, 0, 1.0]])
    P = T.copy(); P[3] = [0.01, -0.02, 0.005, 1.1]           # exercises the perspective divide
    np.savez(os.path.join(OUT, "transform.npz"), pts=pts, T=T, P=P,
             out_T=util.transform_point_cloud(pts, T), out_P=util.transform_point_cloud(pts, P),
             inv_T=util.inverse_transform_point_cloud(util.transform_point_cloud(pts, T), T))

    # ---- BacchusDataset.__getitem__ + collate_fn (blt_dataset.py:173-182,209-271)
    cfg = {"TRAIN": {"AUGMENTATION": False, "BATCH_SIZE": 2}, "MODEL": {"VOXEL_SIZE": 0.1},
           "DATA": {"NUM_WORKER": 0, "SHUFFLE": False}}
    pc_map = np.concatenate([rng.uniform(-3, 3, size=(4000, 3)), rng.uniform(0, 1, size=(4000, 1))], 1)
    scans = []
    for i in range(2):
        idx = rng.choice(len(pc_map), 150, replace=False)
        xyz = pc_map[idx, :3] + rng.normal(0, 0.04, size=(150, 3))
        scans.append(np.concatenate([xyz, rng.uniform(0, 1, size=(150, 1))], 1))
    ds = blt.BacchusDataset(cfg, scans, pc_map)
    items = [ds[i] for i in range(2)]
    batch = blt.BacchusModule.collate_fn(items)
    np.savez(os.path.join(OUT, "bacchus_dataset.npz"), pc_map=pc_map, scan0=scans[0], scan1=scans[1],
             item0=items[0].numpy(), item1=items[1].numpy(), batch=batch.numpy())
    # ---- augmentation (augmentation.py:5-58; BacchusDataset.augment_data, blt_dataset.py:273-278) under fixed seeds
    import sps.datasets.augmentation as aug
    pts32 = torch.tensor(rng.uniform(-10, 10, size=(200, 3)), dtype=torch.float32)
    out = {"pts": pts32.numpy()}
    for seed in (0, 1, 7):
        for name in ("rotate_point_cloud", "rotate_perturbation_point_cloud", "random_flip_point_cloud",
                     "random_scale_point_cloud"):
            torch.manual_seed(seed)
            out[f"{name}_{seed}"] = getattr(aug, name)(pts32.clone()).numpy()
        torch.manual_seed(seed)
        cfg_aug = {"TRAIN": {"AUGMENTATION": True, "BATCH_SIZE": 1}, "MODEL": {"VOXEL_SIZE": 0.1},
                   "DATA": {"NUM_WORKER": 0, "SHUFFLE": False}}
        ds_aug = blt.BacchusDataset(cfg_aug, scans, pc_map, split="train")
        out[f"item_aug_{seed}"] = ds_aug[0].numpy()
    np.savez(os.path.join(OUT, "augmentation.npz"), **out)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
