"""Shared test helpers (no GPU needed to import)."""
from __future__ import annotations

import os

import numpy as np
import torch

CFG = {
    "EXPERIMENT": {"ID": "BLT"},
    "DATA": {"SHUFFLE": False, "NUM_WORKER": 0, "SPLIT": {"TRAIN": ["a"], "VAL": ["b"], "TEST": ["synthetic"]}},
    "TRAIN": {"MAP": "base_map.asc.npy", "BATCH_SIZE": 1, "AUGMENTATION": False, "LR": 7e-5,
              "WEIGHT_DECAY": 1e-4, "LR_EPOCH": 1, "LR_DECAY": 0.99, "MAX_EPOCH": 80},
    "MODEL": {"VOXEL_SIZE": 0.1},
    "FILTER": {"THRESHOLD": 0.84},
}


def state_dict_from_params(params: dict, prefix: str = "model.MinkUNet.") -> dict:
    """oracle parameter dict (reference key names) -> Lightning-style state_dict."""
    sd = {}
    for k, v in params.items():
        sd[prefix + k] = torch.from_numpy(np.ascontiguousarray(v))
    for k in list(sd):
        if k.endswith(".bn.running_var"):
            sd[k.replace("running_var", "num_batches_tracked")] = torch.tensor(0, dtype=torch.long)
    return sd


def net_from_params(params: dict, cfg: dict = CFG):
    from sps_amd.models.models import SPSNet
    net = SPSNet(cfg)
    net.load_state_dict(state_dict_from_params(params))     # strict: same keys as the reference
    return net


LOGIT_EPS = float(np.log(0.84 / 0.16))      # sigmoid(LOGIT_EPS) = 0.84 = FILTER.THRESHOLD


def straddle_params(params: dict, batch: np.ndarray, frac: float = 0.3, gain: float = 8.0,
                    eps: float = 0.84) -> dict:
    """Copy of ``params`` whose `final` layer is rescaled (kernel x gain, bias shifted) so that about ``frac``
    of the scan scores of ``batch`` are >= eps.  With plain Kaiming weights no synthetic score ever reaches
    0.84 (they stay in [0.16, 0.77]) and every label / TP / FP / dIoU comparison would be 0 == 0.
    Uses the C oracle (test infrastructure) to find the logit quantile."""
    from oracle import c_oracle
    _, info = c_oracle.forward(c_oracle.pack_blob(params), batch[:, :5], CFG["MODEL"]["VOXEL_SIZE"], nthreads=8)
    logits = info["logits"][info["inverse"]][batch[:, 4] == 1].astype(np.float64)
    b0 = float(np.asarray(params["final.bias"]).reshape(-1)[0])
    q = float(np.quantile(gain * (logits - b0), 1.0 - frac))
    out = dict(params)
    out["final.kernel"] = (np.asarray(params["final.kernel"]) * np.float32(gain)).astype(np.float32)
    out["final.bias"] = np.full_like(np.asarray(params["final.bias"]), np.log(eps / (1.0 - eps)) - q)
    return out


def plant_threshold_labels(batch: np.ndarray, eps: float = 0.84, every: int = 7) -> np.ndarray:
    """Copy of ``batch`` [N,6] in which every ``every``-th scan row's label is exactly float32(eps), the float32
    just below or the float32 just above it (the reference thresholds with `<` on float32, models.py:97-98, so
    label == eps is class 1 and the neighbour below is class 0)."""
    e = np.float32(eps)
    vals = np.array([e, np.nextafter(e, np.float32(0)), np.nextafter(e, np.float32(1))], np.float32)
    out = batch.copy()
    rows = np.flatnonzero(out[:, 4] == 1)[::every]
    out[rows, 5] = vals[np.arange(len(rows)) % 3]
    return out


def assert_nondegenerate(sums) -> None:
    """[count, TP, FP, FN, TN, ...]: every confusion cell must be populated, else a label comparison is vacuous."""
    s = np.asarray(sums, dtype=np.float64).reshape(-1, 8).sum(axis=0)
    assert s[1] > 0 and s[2] > 0 and s[3] > 0 and s[4] > 0, f"degenerate confusion counts TP/FP/FN/TN = {s[1:5]}"


def write_data_tree(root, n_scans=3, seq="20220629", n_map=3000, n_pts=120, scan_dtype=np.float64):
    """A synthetic $DATA tree in the reference's on-disk layout (blt_dataset.py:49-100)."""
    rng = np.random.default_rng(7)
    os.makedirs(os.path.join(root, "maps"))
    os.makedirs(os.path.join(root, "sequence", seq, "scans"))
    os.makedirs(os.path.join(root, "sequence", seq, "poses"))
    pc_map = np.concatenate([rng.uniform(-4, 4, (n_map, 3)), rng.uniform(0, 1, (n_map, 2))], 1)   # 5 columns: first 4 used
    np.save(os.path.join(root, "maps", "base_map.asc.npy"), pc_map)
    ang = 0.3
    T_map = np.array([[np.cos(ang), -np.sin(ang), 0, 0.5], [np.sin(ang), np.cos(ang), 0, -0.25], [0, 0, 1, 0.1], [0, 0, 0, 1.0]])
    np.savetxt(os.path.join(root, "sequence", seq, "map_transform"), T_map, delimiter=",")
    scans, poses = [], []
    for i in range(n_scans):
        pose = np.eye(4); pose[:3, 3] = [0.2 * i, -0.1 * i, 0.0]
        world = pc_map[rng.choice(len(pc_map), n_pts, replace=False), :3] + rng.normal(0, 0.03, (n_pts, 3))
        sensor = (np.linalg.inv(T_map @ pose) @ np.c_[world, np.ones(len(world))].T).T[:, :3]
        scan = np.c_[sensor, rng.uniform(0, 1, len(sensor))].astype(scan_dtype)
        stamp = f"{1656500000.0 + i:.6f}"
        np.save(os.path.join(root, "sequence", seq, "scans", stamp + ".npy"), scan)
        np.savetxt(os.path.join(root, "sequence", seq, "poses", stamp + ".txt"), pose, delimiter=",")
        scans.append(scan); poses.append(pose)
    return pc_map, T_map, scans, poses
