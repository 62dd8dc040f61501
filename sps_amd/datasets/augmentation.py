"""Training-time point-cloud augmentation -- host-side mirror of the reference's ``sps.datasets.augmentation``
(src/sps/datasets/augmentation.py:5-58), applied by ``BacchusDataset.augment_data`` (blt_dataset.py:273-278) to the
xyz columns of a training item: random yaw, small random rotation about all three axes, random x / y mirror, random
anisotropic scale.  Each function draws from torch's global generator in the reference's order (one ``rand(1)``,
one ``randn(3)``, two ``rand(1)``, one ``rand(1, 3)``), so a seeded run reproduces the reference's augmented clouds
(tests/golden/augmentation.npz)."""
from __future__ import annotations

import torch


def _axis_rotation(axis: int, angle: torch.Tensor) -> torch.Tensor:
    """float32 rotation matrix about coordinate axis 0 (x), 1 (y) or 2 (z)."""
    c, s = float(torch.cos(angle)), float(torch.sin(angle))
    if axis == 0:
        rows = [[1, 0, 0], [0, c, -s], [0, s, c]]
    elif axis == 1:
        rows = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    else:
        rows = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    return torch.tensor(rows, dtype=torch.float32)


def rotate_point_cloud(points: torch.Tensor) -> torch.Tensor:
    """Random rotation about z by an angle uniform in [0, 2 pi) (augmentation.py:5-12): points @ Rz."""
    yaw = torch.rand(1) * 2 * torch.pi
    return points @ _axis_rotation(2, yaw[0]).type_as(points)


def rotate_perturbation_point_cloud(points: torch.Tensor, angle_sigma: float = 0.2, angle_clip: float = 0.5) -> torch.Tensor:
    """Small random rotation (augmentation.py:15-41): angles ~ clip(sigma N(0,1), +-clip) about x, y, z; points @ (Rz Ry Rx)."""
    a = torch.clip(angle_sigma * torch.randn(3), -angle_clip, angle_clip)
    rot = _axis_rotation(2, a[2]) @ _axis_rotation(1, a[1]) @ _axis_rotation(0, a[0])
    return points @ rot.type_as(points)


def random_flip_point_cloud(points: torch.Tensor) -> torch.Tensor:
    """Mirror x and / or y with probability 1/2 each (augmentation.py:44-50); two draws, taken in this order."""
    for column in (0, 1):
        if torch.rand(1).item() > 0.5:
            sign = torch.ones(3)
            sign[column] = -1
            points = points * sign.type_as(points)
    return points


def random_scale_point_cloud(points: torch.Tensor, scale_low: float = 0.8, scale_high: float = 1.2) -> torch.Tensor:
    """Per-axis scale uniform in (scale_low, scale_high] (augmentation.py:53-58)."""
    scales = (scale_low - scale_high) * torch.rand(1, 3) + scale_high
    return points * scales.type_as(points)
