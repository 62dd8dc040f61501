"""Host-side mirror of the reference's online helpers ``sps.datasets.util``
(src/sps/datasets/util.py) for the MI355X path.  ROS message converters (:117-153, :209-232)
are out of scope (no ROS on the box); everything on the per-scan path keeps its name,
argument order, return shapes and assertion messages.
"""
from __future__ import annotations

import os
import time
import weakref

import numpy as np
import torch

from ..models import models

''' Constants (util.py:20-21) '''
SCAN_TIMESTAMP = 1
MAP_TIMESTAMP = 0


class CoordsFeatStruct:
    def __init__(self, cloud_coords, features):
        self.cloud_coords = cloud_coords
        self.features = features


def load_model(cfg=None, weights_pth=None, device="cuda"):
    """util.py:29-46: Lightning checkpoint -> SPSNet on the GPU, eval + frozen.  Keys are renamed
    by stripping "model.MinkUNet."; entries containing "MOSLoss" are dropped."""
    assert cfg != None, "cfg is None!"
    assert weights_pth != None, "weights_pth is None!"
    ckpt = torch.load(weights_pth, map_location="cpu", weights_only=False)
    state_dict = {k.replace("model.MinkUNet.", ""): v for k, v in ckpt["state_dict"].items()}
    state_dict = {k: v for k, v in state_dict.items() if "MOSLoss" not in k}
    model = models.SPSNet(cfg)
    model.model.MinkUNet.load_state_dict(state_dict)
    model = model.to(device)
    model.eval()
    model.freeze()
    return model


def load_point_cloud_map(cfg):
    """util.py:49-64: $DATA/maps/<TRAIN.MAP> (.npy, else text) -> float32 [M, 3] tensor.  The reference logs through
    rospy and calls sys.exit() on failure; without ROS the logging is `print` and the failure an exception."""
    assert cfg != None, "cfg is None!"
    map_id = cfg["TRAIN"]["MAP"]
    map_pth = os.path.join(str(os.environ.get("DATA")), "maps", map_id)
    __, file_extension = os.path.splitext(map_pth)
    print('Loading point cloud map, pth: %s' % (map_pth))
    try:
        point_cloud_map = np.load(map_pth) if file_extension == '.npy' else np.loadtxt(map_pth, dtype=np.float32)
        point_cloud_map = torch.tensor(point_cloud_map[:, :3]).to(torch.float32).reshape(-1, 3)
    except Exception as e:
        raise RuntimeError('Failed to load point cloud map from %s' % map_pth) from e
    print('Point cloud map loaded successfully with %d points' % len(point_cloud_map))
    return point_cloud_map


def to_coords_features(cloud, feature_type='map', ds=0.1, device="cuda"):
    """util.py:67-82: voxel index = (xyz / ds).int()  (float32 division, truncation toward zero)
    and a one-hot source feature ([1,0] scan, [0,1] map)."""
    assert feature_type == 'map' or feature_type == 'scan', "feature_type need to be either 'map' or 'scan'"
    column = 0 if feature_type == 'scan' else 1
    xyz = cloud[:, :3]
    step = torch.tensor([ds, ds, ds], dtype=torch.float32).to(device).type_as(xyz)
    voxels = torch.div(xyz, step).int().to(device)
    onehot = torch.zeros(xyz.shape[0], 2, device=device)
    onehot[:, column] = 1
    return CoordsFeatStruct(voxels, onehot)


_MAP_CACHE = {}


def prune(map_coords_feat=None, scan_coords_feat=None, ds=0.1):
    """util.py:85-114: (unique map voxels) INTERSECT (unique scan voxels) returned as float32
    voxel-corner points ``coords * ds`` plus the number of unique scan voxels.

    The reference re-hashes the whole map through MinkowskiEngine on every call (:86-89); here the
    map's voxel hash stays resident on the device and is rebuilt only when a different map tensor
    is passed.  Output rows are in scan first-occurrence order (the reference's row order is
    unspecified: it is a set)."""
    mc, sc = map_coords_feat.cloud_coords, scan_coords_feat.cloud_coords
    models._require_device_tensor(sc, "scan coordinates")
    models._require_device_tensor(mc, "map coordinates")
    dev = sc.device.index or 0
    with torch.cuda.device(sc.device):
        stream = torch.cuda.current_stream().cuda_stream
        ctx = models.get_context(dev, stream)
        mc32 = mc if (mc.dtype == torch.int32 and mc.stride(1) == 1) else mc.to(torch.int32).contiguous()
        # the uploaded hash belongs to THIS tensor object at THIS version (a different tensor that happens to be
        # allocated at the same address must not hit the cache): weak reference + version counter
        hit = _MAP_CACHE.get(id(ctx))
        if hit is None or hit[0]() is not mc or hit[1] != mc._version:
            ctx.map_upload_voxels(mc32.data_ptr(), mc32.stride(0), mc32.shape[0], stream)
            _MAP_CACHE[id(ctx)] = (weakref.ref(mc), mc._version)
        sc32 = sc if (sc.dtype == torch.int32 and sc.stride(1) == 1) else sc.to(torch.int32).contiguous()
        n = sc32.shape[0]
        out = torch.empty((n, 3), dtype=torch.float32, device=sc.device)
        n_sub, n_scan_vox = ctx.submap_voxel_ijk(sc32.data_ptr(), sc32.stride(0) if n else 3, n, float(ds),
                                                 out.data_ptr(), stream)
    return out[:n_sub], n_scan_vox


def add_timestamp(data, stamp, device):
    """util.py:156-160: append a constant time column."""
    column = torch.full((len(data), 1), stamp, dtype=data.dtype, device=device)
    return torch.hstack([data.to(device), column])


def infer(scan_points, submap_points, model, device="cuda"):
    """util.py:163-184: [b=0 | scan xyz, t=1 ; submap xyz, t=0] -> model -> scores of the scan rows.

    Stream-ordered, no host synchronisation.  Contract for unrepresentable coordinates (outside the 64-bit voxel key,
    include/sps_hip.h): their scores are NaN and the context's sticky error flag is set; it surfaces as SpsError
    (SPS_ERR_RANGE) from the next synchronising call -- ``models.get_context(...).check_errors(stream)``,
    ``SPSNet.step_metrics`` or ``StableFilter`` results -- never silently."""
    start_time = time.time()
    assert scan_points.size(-1) == 3, f"Expected 3 columns, but the scan tensor has {scan_points.size(-1)} columns."
    n_scan = len(scan_points)
    assert submap_points.size(-1) == 3, f"Expected 3 columns, but the submap tensor has {submap_points.size(-1)} columns."
    n = n_scan + len(submap_points)
    tensor = torch.empty((n, 5), dtype=scan_points.dtype, device=device)
    tensor[:, 0] = 0
    tensor[:n_scan, 1:4] = scan_points
    tensor[:n_scan, 4] = SCAN_TIMESTAMP
    tensor[n_scan:, 1:4] = submap_points
    tensor[n_scan:, 4] = MAP_TIMESTAMP
    with torch.no_grad():
        scores = model.forward(tensor)
    scan_scores = scores[:n_scan]
    elapsed_time = time.time() - start_time
    return scan_scores.to(device), elapsed_time


def transform_point_cloud(point_cloud, transformation_matrix):
    """util.py:187-194: p' = T [p;1] with perspective divide (float64 numpy)."""
    h = np.concatenate([point_cloud, np.ones((point_cloud.shape[0], 1))], axis=1)
    t = np.dot(h, transformation_matrix.T)
    return t[:, :3] / t[:, 3][:, np.newaxis]


def inverse_transform_point_cloud(transformed_point_cloud, transformation_matrix):
    """util.py:197-206."""
    return transform_point_cloud(transformed_point_cloud, np.linalg.inv(transformation_matrix))


def calculate_metrics(true_labels, predicted_labels):
    """util.py:285-299: positive class = 1 (unstable); zero guards on precision/recall/f1 only
    (accuracy and dIoU divide unguarded, NaN on an empty denominator as in the reference)."""
    t = np.asarray(true_labels)
    p = np.asarray(predicted_labels)
    tp = np.sum((t == 1) & (p == 1))
    tn = np.sum((t == 0) & (p == 0))
    fp = np.sum((t == 0) & (p == 1))
    fn = np.sum((t == 1) & (p == 0))
    precision = tp / (tp + fp) if (tp + fp) != 0 else 0
    recall = tp / (tp + fn) if (tp + fn) != 0 else 0
    f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) != 0 else 0
    accuracy = (tp + tn) / (tp + tn + fp + fn)
    dIoU = tp / (tp + fn + fp)
    return precision, recall, f1, accuracy, dIoU
