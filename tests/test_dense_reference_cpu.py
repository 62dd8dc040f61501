"""The independently written dense-torch network (tests/dense_reference.py: F.conv3d / F.conv_transpose3d / F.batch_norm on a
zero-filled grid, re-derived from the reference's minkunet.py / resnet.py / models.py) against the two oracles, END TO END:
pins the oracles' wiring -- layer order, the four concatenations' operand order (minkunet.py:192,200,208,216), the residual /
downsample branch (resnet.py:98-108), stride and transposed geometry -- against code that shares nothing with them.  (The GPU
counterpart compares the HIP path with the same reference directly: tests/test_hip_dense_reference.py.)"""
import numpy as np
import torch

from oracle import c_oracle, sps_oracle as O
from sps_amd import synthetic
from tests.dense_reference import DenseSPS
from tests.helpers import state_dict_from_params

VS = 0.1
TAP_STRIDE = {"out_p1": 1, "block1": 2, "block2": 4, "block3": 8, "block4": 16, "block5": 8, "block6": 4, "block7": 2, "block8": 1}


def test_dense_torch_network_equals_the_oracles_end_to_end():
    batch = synthetic.small_scene(seed=21, n_scan=1500, extent=4.0)
    batch[:, 1:4] -= 2.3                                          # negative octants: floor strides, re-quantised corners
    params = O.random_params(seed=4)
    ref, info = O.sps_forward(params, batch[:, :5], VS, keep=True)
    dense = DenseSPS(state_dict_from_params(params), dtype=torch.float64)
    taps = {}
    scores, logits = dense.forward(torch.from_numpy(batch), VS, taps=taps)
    np.testing.assert_allclose(logits.numpy(), info["logits"][info["inverse"]], rtol=0, atol=5e-5)
    np.testing.assert_allclose(scores.numpy(), ref, rtol=0, atol=2e-6)
    cm = info["cm"]
    taps["out_p1"] = taps.pop("conv0")
    for name, want in info["inter"].items():                      # every tapped feature map, row by row at the oracle's coordinates
        ts = TAP_STRIDE[name]
        got = DenseSPS.rows_at(taps[name], taps["_origin"], cm.coords[ts], ts).numpy()
        assert got.shape == want.shape, name
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4, err_msg=name)
        # ... and the dense grid is zero everywhere else: the coordinate sets (stride pyramid, transposed outputs) agree
        assert int((taps[name].abs().sum(2) > 0).sum()) <= len(want), name
    # the C restatement (the CPU baseline) on the same input
    ref_c, info_c = c_oracle.forward(c_oracle.pack_blob(params), batch[:, :5], VS, nthreads=4)
    np.testing.assert_allclose(scores.numpy(), ref_c, rtol=0, atol=2e-6)


def test_dense_reference_concat_order_and_residual_are_observable():
    """The check has teeth: swapping the operands of one concatenation, or dropping one block's downsample branch, in the
    dense network moves the logits far beyond the comparison's tolerance."""
    batch = synthetic.small_scene(seed=22, n_scan=800, extent=3.0)
    params = O.random_params(seed=5)
    _, info = O.sps_forward(params, batch[:, :5], VS)
    want = info["logits"][info["inverse"]]
    sd = state_dict_from_params(params)

    class SwappedCat(DenseSPS):
        def block(self, x, name, mask):
            if name == "block6":                                  # [skip, up] instead of [up, skip] (minkunet.py:200)
                x = torch.cat((x[:, :, 32:], x[:, :, :32]), 2)
            return super().block(x, name, mask)

    class NoDownsample(DenseSPS):
        def block(self, x, name, mask):
            if name != "block3":
                return super().block(x, name, mask)
            y = torch.relu(self.bn(self.conv81(x, name + ".0.conv1", mask), name + ".0.norm1", mask))
            y = self.bn(self.conv81(y, name + ".0.conv2", mask), name + ".0.norm2", mask)
            return torch.relu(y)

    for cls in (SwappedCat, NoDownsample):
        _, logits = cls(sd, dtype=torch.float64).forward(torch.from_numpy(batch), VS)
        assert float(np.abs(logits.numpy() - want).max()) > 1e-2, cls.__name__
