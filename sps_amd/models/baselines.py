"""The reference's two baseline networks on the MI355X backbone.

  MOS4DNet(voxel_size).forward(coordinates[N,5]) -> logits[N]          reference c_ws/src/mos4d/scripts/mos4d.py:11-32
  MapMOSNet(voxel_size).forward(coordinates[N,5], indices[N,1]) -> logits[N]
           .predict(scan_input, map_input, scan_indices, map_indices) -> (logits_scan, logits_map)
           .to_label(logits)                                           reference c_ws/src/mapmos/scripts/mapmos.py:32-89

Both wrap the same CustomMinkUNet14 wiring as SPS (c_ws/src/mapmos/scripts/minkunet.py:84-308 is the same
layer list as src/sps/models/MinkowskiEngine/minkunet.py:52-219 with the SPS widths), so they run through the
same HIP kernels; what differs is the `final` width (3 for 4DMOS), the input feature (MapMOS: per-point
``1 + (i_max - i) / (i_max - i_min)``, voxel feature = mean over the voxel's points) and that raw logits
are returned (no sigmoid).  ``self.MinkUNet`` holds the parameters under the reference's state_dict keys
(the nodes load ``{k.replace("model.MinkUNet.", ""): v}`` / ``"mos.MinkUNet."`` into it:
mos4d_node.py:63-70, mapmos_node.py:46-54).  No CPU fallback: a CPU tensor raises.
"""
from __future__ import annotations

import copy
import math

import torch

from .models import NativeBackboneModule, get_context

T_MIN, T_MAX = -16, 15          # include/sps_hip.h SPS_T_MIN / SPS_T_MAX


def _t_base(coordinates: torch.Tensor) -> float:
    """Integer shift that brings floor(t) into the native key range.  The 4DMOS node numbers scans with an
    ever-growing index (mos4d_node.py:98-104); every stride in the network is [2,2,2,1], so the result
    does not depend on a common shift of t.  One device->host read (the nodes sync right after anyway)."""
    if coordinates.shape[0] == 0:
        return 0.0
    lo = math.floor(float(coordinates[:, 4].min()))
    hi = math.floor(float(coordinates[:, 4].max()))
    if T_MIN <= lo and hi <= T_MAX:
        return 0.0
    return float(lo)            # hi - lo > 15 is reported by the native range check


class _HeadModule(NativeBackboneModule):
    def _run(self, coordinates: torch.Tensor, features, voxel_size: float, activation: int = 0) -> torch.Tensor:
        coordinates = self._prepare_coordinates(coordinates)
        n = coordinates.shape[0]
        oc = self.MinkUNet.out_channels
        if features is not None:
            features = features.reshape(-1).to(device=coordinates.device, dtype=torch.float32).contiguous()
            if features.numel() != n:
                raise ValueError(f"{features.numel()} features for {n} coordinates")
        with torch.cuda.device(coordinates.device):
            stream = torch.cuda.current_stream().cuda_stream
            ctx = get_context(coordinates.device.index or 0, stream)
            self._sync_weights(ctx)
            out = torch.empty((n, oc), dtype=torch.float32, device=coordinates.device)
            ctx.forward_head(coordinates.data_ptr(), coordinates.stride(0) if n else 5, n, voxel_size,
                             features.data_ptr() if features is not None and n else None, _t_base(coordinates),
                             out.data_ptr(), oc, activation, stream)
        return out

    def freeze(self):
        for p in self.parameters():
            p.requires_grad_(False)
        return self.eval()


class MOS4DNet(_HeadModule):
    def __init__(self, voxel_size):
        super().__init__()
        self.ds = voxel_size
        self._init_backbone(out_channels=3)

    def forward(self, coordinates: torch.Tensor) -> torch.Tensor:
        out = self._run(coordinates, None, float(self.ds))
        return out[:, 2].reshape(-1)                                   # mos4d.py:32


class MapMOSNet(_HeadModule):
    def __init__(self, voxel_size: float):
        super().__init__()
        self.voxel_size = voxel_size
        self._init_backbone(out_channels=1)

    def predict(self, scan_input, map_input, scan_indices, map_indices):
        # [batch_idx = 0, x, y, z, t] with t = 0 for the scan and -1 for the map (mapmos.py:39-47), written
        # straight into one buffer; scan rows come first, so the reference's `t == 0` mask is a prefix
        scan_input, map_input = scan_input.reshape(-1, 3), map_input.reshape(-1, 3)
        ns, nm = scan_input.shape[0], map_input.shape[0]
        coordinates = torch.zeros((ns + nm, 5), dtype=scan_input.dtype, device=scan_input.device)
        coordinates[:ns, 1:4] = scan_input
        coordinates[ns:, 1:4] = map_input
        coordinates[ns:, 4] = -1
        indices = torch.cat([scan_indices.reshape(-1, 1), map_indices.reshape(-1, 1)])
        logits = self.forward(coordinates, indices)
        return logits[:ns], logits[ns:]

    def forward(self, coordinates: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        # normalise indices (mapmos.py:65-71); the division by the voxel size happens in the native quantiser
        i_max = torch.max(indices)
        i_min = torch.min(indices)
        if i_min == i_max:
            features = 1.0 * torch.ones_like(indices)
        else:
            features = 1 + (i_max - indices) / (i_max - i_min)
        return self._run(coordinates.reshape(-1, 5), features, float(self.voxel_size)).reshape(-1)

    def to_label(self, logits):
        labels = copy.deepcopy(logits)
        mask = logits > 0
        labels[mask] = 1.0
        labels[~mask] = 0.0
        return labels
