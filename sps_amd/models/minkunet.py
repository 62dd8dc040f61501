"""Parameter container for the SPS backbone (CustomMinkUNet = MinkUNet14 wiring with
PLANES=(8,16,32,64,64,32,16,8), INIT_DIM=8, D=4).

The reference builds this network out of MinkowskiEngine modules
(src/sps/models/MinkowskiEngine/minkunet.py:52-159, resnet.py:96-126, customminkunet.py:10-12,
BasicBlock body spelled out at c_ws/src/mapmos/scripts/minkunet.py:31-82).  Here the module tree
only HOLDS the parameters -- with exactly the reference's state_dict keys and shapes
(SURVEY.md App. B), so Lightning checkpoints load with ``strict=True`` -- while the
arithmetic runs in libsps_hip.so.  The layer order / blob layout is owned by the native library
(``_native.weight_layout()``), not duplicated here.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

PLANES = (8, 16, 32, 64, 64, 32, 16, 8)
INIT_DIM = 8


class SparseConvParams(nn.Module):
    """Stands for ME.MinkowskiConvolution / MinkowskiConvolutionTranspose: ``kernel`` is
    [K, C_in, C_out] (2-D [C_in, C_out] when K == 1), optional ``bias`` [1, C_out]."""

    def __init__(self, volume: int, cin: int, cout: int, bias: bool = False, transpose: bool = False):
        super().__init__()
        self.volume, self.cin, self.cout, self.transpose = volume, cin, cout, transpose
        shape = (cin, cout) if volume == 1 else (volume, cin, cout)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty(1, cout, dtype=torch.float32)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # ME default: uniform(-1/sqrt(n), 1/sqrt(n)), n = (C_out if transpose else C_in) * volume
        n = (self.cout if self.transpose else self.cin) * self.volume
        bound = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def kaiming_normal_fan_out(self):
        # ME.utils.kaiming_normal_(kernel, mode="fan_out", nonlinearity="relu") -- resnet.py:90.
        # 3-D kernels: fan_out = C_out * volume; 2-D kernels follow the Linear convention
        # (fan_out = size(0)).
        fan = self.kernel.shape[0] if self.kernel.dim() == 2 else self.cout * self.volume
        with torch.no_grad():
            self.kernel.normal_(0.0, math.sqrt(2.0 / fan))


class BatchNormParams(nn.Module):
    """Stands for ME.MinkowskiBatchNorm: wraps ``self.bn = nn.BatchNorm1d(C)`` (resnet.py:93-94)."""

    def __init__(self, channels: int):
        super().__init__()
        self.bn = nn.BatchNorm1d(channels)


class BasicBlockParams(nn.Module):
    def __init__(self, inplanes: int, planes: int, downsample: nn.Module | None):
        super().__init__()
        self.conv1 = SparseConvParams(81, inplanes, planes)
        self.norm1 = BatchNormParams(planes)
        self.conv2 = SparseConvParams(81, planes, planes)
        self.norm2 = BatchNormParams(planes)
        self.downsample = downsample


class CustomMinkUNet(nn.Module):
    def __init__(self, in_channels: int = 1, out_channels: int = 1, D: int = 4):
        super().__init__()
        if in_channels != 1 or D != 4 or not 1 <= out_channels <= 8:
            raise NotImplementedError("the MI355X path implements CustomMinkUNet(in_channels=1, out_channels<=8, D=4): "
                                      "SPS / MapMOS (out_channels=1, models.py:17, mapmos.py:36) and 4DMOS "
                                      "(out_channels=3, mos4d.py:15)")
        self.D = D
        self.out_channels = out_channels
        self.inplanes = INIT_DIM
        self.conv0p1s1 = SparseConvParams(125, in_channels, self.inplanes)
        self.bn0 = BatchNormParams(self.inplanes)
        for i, name in enumerate(("conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2")):
            setattr(self, name, SparseConvParams(8, self.inplanes, self.inplanes))
            setattr(self, f"bn{i + 1}", BatchNormParams(self.inplanes))
            setattr(self, f"block{i + 1}", self._make_layer(PLANES[i]))
        skips = (PLANES[2], PLANES[1], PLANES[0], INIT_DIM)
        for i, name in enumerate(("convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2")):
            setattr(self, name, SparseConvParams(8, self.inplanes, PLANES[4 + i], transpose=True))
            setattr(self, f"bntr{4 + i}", BatchNormParams(PLANES[4 + i]))
            self.inplanes = PLANES[4 + i] + skips[i]
            setattr(self, f"block{5 + i}", self._make_layer(PLANES[4 + i]))
        self.final = SparseConvParams(1, PLANES[7], out_channels, bias=True)
        self.weight_initialization()

    def _make_layer(self, planes: int) -> nn.Sequential:
        downsample = None
        if self.inplanes != planes:                                    # resnet.py:98
            downsample = nn.Sequential(SparseConvParams(1, self.inplanes, planes), BatchNormParams(planes))
        block = BasicBlockParams(self.inplanes, planes, downsample)
        self.inplanes = planes
        return nn.Sequential(block)

    def weight_initialization(self):
        """resnet.py:87-94: Kaiming fan_out on every (non-transposed) conv, BN gamma=1, beta=0."""
        for m in self.modules():
            if isinstance(m, SparseConvParams) and not m.transpose:
                m.kaiming_normal_fan_out()
            if isinstance(m, BatchNormParams):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def forward(self, x):
        raise RuntimeError("CustomMinkUNet holds parameters only; the sparse convolutions run inside "
                           "SPSModel.forward through libsps_hip.so")
