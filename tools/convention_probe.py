#!/usr/bin/env python3
"""Which MinkowskiEngine conventions does a checkpoint follow?

    python tools/convention_probe.py -w 420_601.ckpt -c config/config.yaml --scan batch.npy [--top 8]

``batch.npy`` is one collated item [N, 6] = (b, x, y, z, t, label) as ``BacchusDataset.__getitem__`` + ``collate_fn``
produce it (/root/reference/src/sps/datasets/blt_dataset.py:173-244), labels = ground-truth stability.  The checkpoint is
run through the HIP path once per combination of the options in sps_amd/conventions.py (32); a trained network read
with the wrong kernel-index convention is a scrambled network, so the right combination stands out by R2 / loss / uIoU
against the labels.  Prints one line per combination, best R2 first, and the line to put into
``sps_amd/conventions.py: DEFAULT`` (or ``MODEL.ME_CONVENTIONS`` of the config) if the winner is not the default.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sps_amd import conventions as CV  # noqa: E402
from sps_amd.models.models import SPSNet  # noqa: E402


def probe(state_dict: dict, cfg: dict, batch: torch.Tensor, combos=None):
    """[(conventions, metrics dict)] sorted by R2 (NaN last), best first.  ``batch`` is a device tensor [N, 6]."""
    net = SPSNet(cfg)
    combos = list(combos or CV.all_combinations())
    try:
        net.load_state_dict(state_dict)
    except RuntimeError as e:
        # a non-square 1x1 kernel stored [C_out, C_in] fails the strict load under the default reading (the load hook only
        # accepts that shape once lin_layout = out_in is declared): load under out_in and probe that half of the combinations
        if "size mismatch" not in str(e):
            raise
        net = SPSNet(cfg)
        net.model.set_me_conventions(CV.parse("lin_layout=out_in"))
        net.load_state_dict(state_dict)
        combos = [cv for cv in combos if cv.lin_layout == "out_in"]
        print("the checkpoint's 1x1 kernels are stored [C_out, C_in]: probing the lin_layout = out_in combinations only", file=sys.stderr)
    net = net.cuda().eval().freeze()
    out = []
    for cv in combos:
        net.model.set_me_conventions(cv)
        net.predict_loss.clear()
        m = net.predict_step(batch, 0)
        out.append((cv, m))
    out.sort(key=lambda e: -e[1]["r2"] if np.isfinite(e[1]["r2"]) else np.inf)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-w", "--weights", required=True, help="Lightning checkpoint (predict.py -w)")
    ap.add_argument("-c", "--config", default=os.path.join(os.path.dirname(__file__), "..", "config", "config.yaml"))
    ap.add_argument("--scan", required=True, help=".npy with one collated item [N,6] = (b,x,y,z,t,label)")
    ap.add_argument("--top", type=int, default=32)
    a = ap.parse_args()
    cfg = yaml.safe_load(open(a.config))
    ckpt = torch.load(a.weights, map_location="cpu", weights_only=False)
    sd = {k: v for k, v in ckpt["state_dict"].items() if "MOSLoss" not in k}
    batch = torch.from_numpy(np.load(a.scan).astype(np.float32)).cuda()
    res = probe(sd, cfg, batch)
    print(f"{'R2':>9} {'loss':>9} {'uIoU':>7} {'F1':>7}  conventions")
    for cv, m in res[: a.top]:
        print(f"{m['r2']:9.4f} {m['loss']:9.5f} {m['dIoU']:7.4f} {m['f1']:7.4f}  {cv.describe()}{'   <- default' if cv.is_default else ''}")
    best = res[0][0]
    if best.is_default:
        print("the default conventions explain this checkpoint best")
    else:
        print(f"best: MODEL.ME_CONVENTIONS: \"{best.describe()}\"   (or DEFAULT = MEConventions(...) in sps_amd/conventions.py)")


if __name__ == "__main__":
    main()
