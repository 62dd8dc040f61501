"""CustomMinkUNet14 + the SPS head written with DENSE torch ops on a zero-filled grid -- a second fp32 reference for the
WIRING of the network that shares no code with `oracle/` and none with the HIP path (test infrastructure only).

What it restates, from the reference's sources alone:
  * `SPSModel.forward` (src/sps/models/models.py:20-30): coordinates / [1, vs, vs, vs, 1] in float32, constant feature 0.5,
    sparse voxelisation (floor, one voxel per distinct (b, x, y, z, t)), backbone, slice back to the points, sigmoid;
  * `MinkUNetBase.forward` (src/sps/models/MinkowskiEngine/minkunet.py:161-219): the layer order, the four `ME.cat`s with the
    up-sampled tensor FIRST and the encoder's skip SECOND (:192, :200, :208, :216), channel widths of `CustomMinkUNet14`
    (customminkunet.py:10-12), `final` with bias (:152-158);
  * `BasicBlock.forward` (c_ws/src/mapmos/scripts/minkunet.py:65-82; built by resnet.py:96-126): conv-bn-relu-conv-bn, the
    residual through `downsample` = 1x1 conv + BN iff the channel count changes (resnet.py:98-108), add, relu.

How a sparse layer becomes a dense one: a sparse tensor is a dense grid [B, T, C, Z, Y, X] that is ZERO off the active
sites; a generalised sparse convolution (out[u] += in[u + o_k] @ W[k] for every active input u + o_k) is
then `F.conv3d` per time slice (the time taps of the 3x3x3x3 kernel: one conv3d per dt), followed by a multiplication with
the OUTPUT level's occupancy mask (sparse outputs exist only on the output coordinate set).  Stride-2 layers
(kernel [2,2,2,1], App. A.9) are `F.conv3d(stride=2)`, their coordinate set is the max-pool of the finer mask; transposed
layers (App. A.10) are `F.conv_transpose3d(stride=2)` masked with the ENCODER's mask of that level.  The grid origin is a
multiple of 16 voxels, so floor-division by the strides agrees with the sparse coordinates' floors.  Eval-mode BatchNorm is
`F.batch_norm(training=False)` followed by the mask (off-site values would otherwise become `shift`).

The ME conventions (kernel-offset enumeration x fastest, even kernels starting at 0, transposed map = stride map with in / out
swapped, 2-D kernel for kernel_size = 1: SURVEY App. A items 6-11) enter only in how a `[K, C_in, C_out]` tensor is reshaped
into a conv3d weight -- they are the repository's stated assumption, here as in the oracle; everything else (layer order, concat
order, residual wiring, masks, BN placement) is re-derived from the reference text.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

PLANES = (8, 16, 32, 64, 64, 32, 16, 8)     # customminkunet.py:11
INIT_DIM = 8                                  # customminkunet.py:12


class DenseSPS:
    def __init__(self, state_dict, prefix="model.MinkUNet.", device="cpu", dtype=torch.float32):
        self.dev, self.dt = torch.device(device), dtype
        self.p = {k[len(prefix):]: v.detach().to(self.dev, dtype if v.is_floating_point() else v.dtype)
                  for k, v in state_dict.items() if k.startswith(prefix)}

    # ---- layers ------------------------------------------------------------------------------------------------------
    def bn(self, x, name, mask):
        p = self.p
        B, T = x.shape[:2]
        y = F.batch_norm(x.flatten(0, 1), p[name + ".bn.running_mean"], p[name + ".bn.running_var"], p[name + ".bn.weight"],
                         p[name + ".bn.bias"], False, 0.0, 1e-5)
        return y.unflatten(0, (B, T)) * mask

    def conv_space(self, x, name, ks, mask):
        """kernel [ks, ks, ks, 1], stride 1, odd ks: one conv3d per time slice; W[k], k = ix + ks (iy + ks iz)."""
        W = self.p[name + ".kernel"]
        cin, cout = W.shape[1:]
        w = W.reshape(ks, ks, ks, cin, cout).permute(4, 3, 0, 1, 2).contiguous()          # [co][ci][z][y][x]
        B, T = x.shape[:2]
        return F.conv3d(x.flatten(0, 1), w, padding=ks // 2).unflatten(0, (B, T)) * mask

    def conv81(self, x, name, mask):
        """kernel 3 on all four axes: W[k], k = (dx+1) + 3 (dy+1) + 9 (dz+1) + 27 (dt+1); in = out + offset."""
        W = self.p[name + ".kernel"]
        cin, cout = W.shape[1:]
        w = W.reshape(3, 3, 3, 3, cin, cout).permute(0, 5, 4, 1, 2, 3).contiguous()       # [dt][co][ci][dz][dy][dx]
        T = x.shape[1]
        out = []
        for to in range(T):
            acc = None
            for dt in (-1, 0, 1):
                if 0 <= to + dt < T:
                    y = F.conv3d(x[:, to + dt], w[dt + 1], padding=1)
                    acc = y if acc is None else acc + y
            out.append(acc)
        return torch.stack(out, 1) * mask

    def conv_down(self, x, name, mask_coarse):
        """kernel [2,2,2,1], stride [2,2,2,1]: out[u] = sum_o in[2u + o] @ W[k], k = ox + 2 oy + 4 oz, o in {0,1}^3."""
        W = self.p[name + ".kernel"]
        cin, cout = W.shape[1:]
        w = W.reshape(2, 2, 2, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
        B, T = x.shape[:2]
        return F.conv3d(x.flatten(0, 1), w, stride=2).unflatten(0, (B, T)) * mask_coarse

    def conv_up(self, x, name, mask_fine):
        """transposed, kernel [2,2,2,1], stride 2: fine voxel v = 2u + o receives in[u] @ W[k(o)]; only existing fine voxels."""
        W = self.p[name + ".kernel"]
        cin, cout = W.shape[1:]
        w = W.reshape(2, 2, 2, cin, cout).permute(3, 4, 0, 1, 2).contiguous()             # [ci][co][z][y][x]
        B, T = x.shape[:2]
        return F.conv_transpose3d(x.flatten(0, 1), w, stride=2).unflatten(0, (B, T)) * mask_fine

    def conv1x1(self, x, W, mask, bias=None):
        y = torch.einsum("btczyx,cd->btdzyx", x, W.reshape(W.shape[-2], W.shape[-1]))
        if bias is not None:
            y = y + bias.reshape(1, 1, -1, 1, 1, 1)
        return y * mask

    def block(self, x, name, mask):
        """BasicBlock: relu(bn2(conv2(relu(bn1(conv1(x))))) + residual), residual = downsample(x) when present."""
        y = torch.relu(self.bn(self.conv81(x, name + ".0.conv1", mask), name + ".0.norm1", mask))
        y = self.bn(self.conv81(y, name + ".0.conv2", mask), name + ".0.norm2", mask)
        if name + ".0.downsample.0.kernel" in self.p:
            r = self.bn(self.conv1x1(x, self.p[name + ".0.downsample.0.kernel"], mask), name + ".0.downsample.1", mask)
        else:
            r = x
        return torch.relu(y + r)

    # ---- the path ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, points, voxel_size, taps=None):
        """points [N, >= 5] = (b, x, y, z, t) float32 -> (scores [N], logits [N]); `taps` (a dict) receives, per tapped name,
        the feature grid and a function that reads rows at integer coordinates."""
        pts = points[:, :5].to(self.dev, torch.float32)
        quant = torch.tensor([1.0, voxel_size, voxel_size, voxel_size, 1.0], dtype=torch.float32, device=self.dev)
        q = torch.floor(pts / quant).long()                                                   # models.py:21 + ME floor
        b, x, y, z, t = q.unbind(1)
        xyz = torch.stack([x, y, z], 1)
        lo = torch.div(xyz.min(0).values, 16, rounding_mode="floor") * 16 - 16                # multiple of 16, one coarse cell spare
        hi = xyz.max(0).values + 16
        size = (torch.div(hi - lo + 15, 16, rounding_mode="floor") * 16).tolist()             # (X, Y, Z), multiples of 16
        b0, t0 = int(b.min()), int(t.min())
        B, T = int(b.max()) - b0 + 1, int(t.max()) - t0 + 1
        m0 = torch.zeros((B, T, 1, size[2], size[1], size[0]), dtype=self.dt, device=self.dev)
        ix = (b - b0, t - t0, torch.zeros_like(b), z - lo[2], y - lo[1], x - lo[0])
        m0[ix] = 1.0
        masks = [m0]
        for _ in range(4):                                                                    # coordinate sets at strides 2..16
            mm = masks[-1]
            masks.append(F.max_pool3d(mm.flatten(0, 1), 2).unflatten(0, (B, T)))
        m1, m2, m4, m8, m16 = masks
        relu = torch.relu

        def tap(name, val):
            if taps is not None:
                taps[name] = val

        x0 = 0.5 * m0                                                                         # models.py:22
        out_p1 = relu(self.bn(self.conv_space(x0, "conv0p1s1", 5, m1), "bn0", m1))
        tap("conv0", out_p1)
        out = relu(self.bn(self.conv_down(out_p1, "conv1p1s2", m2), "bn1", m2))
        out_b1p2 = self.block(out, "block1", m2)
        tap("block1", out_b1p2)
        out = relu(self.bn(self.conv_down(out_b1p2, "conv2p2s2", m4), "bn2", m4))
        out_b2p4 = self.block(out, "block2", m4)
        tap("block2", out_b2p4)
        out = relu(self.bn(self.conv_down(out_b2p4, "conv3p4s2", m8), "bn3", m8))
        out_b3p8 = self.block(out, "block3", m8)
        tap("block3", out_b3p8)
        out = relu(self.bn(self.conv_down(out_b3p8, "conv4p8s2", m16), "bn4", m16))
        out = self.block(out, "block4", m16)
        tap("block4", out)
        out = relu(self.bn(self.conv_up(out, "convtr4p16s2", m8), "bntr4", m8))
        out = self.block(torch.cat((out, out_b3p8), 2), "block5", m8)                         # minkunet.py:192
        tap("block5", out)
        out = relu(self.bn(self.conv_up(out, "convtr5p8s2", m4), "bntr5", m4))
        out = self.block(torch.cat((out, out_b2p4), 2), "block6", m4)                         # :200
        tap("block6", out)
        out = relu(self.bn(self.conv_up(out, "convtr6p4s2", m2), "bntr6", m2))
        out = self.block(torch.cat((out, out_b1p2), 2), "block7", m2)                         # :208
        tap("block7", out)
        out = relu(self.bn(self.conv_up(out, "convtr7p2s2", m1), "bntr7", m1))
        out = self.block(torch.cat((out, out_p1), 2), "block8", m1)                           # :216
        tap("block8", out)
        logit_grid = self.conv1x1(out, self.p["final.kernel"], m1, self.p["final.bias"])     # minkunet.py:219
        logits = logit_grid[ix]                                                               # slice (models.py:28)
        if taps is not None:
            taps["_origin"] = (b0, t0, lo)
        return torch.sigmoid(logits), logits

    @staticmethod
    def rows_at(grid, origin, coords, stride):
        """Feature rows of `grid` (a tapped [B, T, C, Z, Y, X] tensor at tensor stride `stride`) at integer voxel coordinates
        [n, 5] = (b, x, y, z, t) given in units of the level-0 voxel (ME convention: multiples of `stride`)."""
        b0, t0, lo = origin
        c = torch.as_tensor(coords, device=grid.device).long()
        zi = torch.div(c[:, 3] - lo[2], stride, rounding_mode="floor")
        yi = torch.div(c[:, 2] - lo[1], stride, rounding_mode="floor")
        xi = torch.div(c[:, 1] - lo[0], stride, rounding_mode="floor")
        return grid[c[:, 0] - b0, c[:, 4] - t0, :, zi, yi, xi]
