"""THE switch-point for every MinkowskiEngine convention a real checkpoint could contradict.

The arithmetic of the reference's network lives in NVIDIA/MinkowskiEngine, cloned un-pinned from master
(/root/reference/Dockerfile:38-40) and absent here; the pretrained ``420_601.ckpt`` (/root/reference/Readme.md:104-107)
stores its convolution kernels ``[K, C_in, C_out]`` indexed by ME's kernel-offset enumeration
(layers: /root/reference/src/sps/models/MinkowskiEngine/minkunet.py:55-146).  SURVEY.md Appendix A states those
conventions from knowledge of ME 0.5.x (items marked with a diamond there); nothing in this repository can prove them.

Every one of them changes only WHICH weight slice ``W[k]`` meets WHICH geometric kernel offset (or how a 2-D kernel is
stored) -- never the coordinate sets, the kernel maps or the kernels.  So the HIP library always computes in ONE internal
(canonical) convention, and a checkpoint written under another one is brought to it by an explicit index permutation of
the flat weight blob, applied exactly where the blob is packed (``NativeBackboneModule.device_weights`` for inference,
``_TrainForward`` for the training step, which also routes the gradients back through the inverse permutation).
If the real checkpoint meets the real ME and a guess turns out wrong, the fix is ONE line: ``DEFAULT`` below
(``tools/convention_probe.py`` ranks all combinations by R2 / uIoU on a labelled scan to find the right one).

Canonical convention (= ``MEConventions()``):
  * kernel offsets enumerated with the first spatial axis (x) fastest and t slowest: k = ix + kx (iy + ky (iz + kz it));
  * odd kernel sizes centred, ``in = out + o_k`` (cross-correlation);
  * even kernel sizes (the [2,2,2,1] stride-2 kernels) start at 0: index k <-> offset (dx, dy, dz) in {0, s}^3 ascending;
  * a transposed convolution uses the forward stride map with in / out swapped and the SAME kernel index;
  * ``kernel_size = 1`` kernels are 2-D ``[C_in, C_out]`` (a 3-D ``[1, C_in, C_out]`` tensor is accepted as well).

``oracle/sps_oracle.py`` takes the same options and implements them *geometrically* (in the offsets it enumerates),
independently of the permutation computed here; ``tests/test_conventions.py`` (CPU) and ``tests/test_hip_conventions.py``
(GPU) hold the two against each other.
"""
from __future__ import annotations

import itertools
from dataclasses import asdict, dataclass, fields, replace

import numpy as np

OPTIONS = {
    "offset_order": ("x_fastest", "t_fastest"),        # App. A.7
    "odd_kernel_sign": ("plus", "minus"),              # App. A.8: in = out + o_k | in = out - o_k (mirrored index)
    "even_kernel_order": ("ascending", "descending"),  # App. A.6/A.9: k <-> {0, +s} | k <-> {+s, 0} per axis (k | 7 - k)
    "transpose_index": ("same", "mirrored"),           # App. A.10: W[k] | W[7 - k] for the transposed convolutions
    "lin_layout": ("in_out", "out_in"),                # App. A.11: 2-D kernel [C_in, C_out] | [C_out, C_in]
}


@dataclass(frozen=True)
class MEConventions:
    offset_order: str = "x_fastest"
    odd_kernel_sign: str = "plus"
    even_kernel_order: str = "ascending"
    transpose_index: str = "same"
    lin_layout: str = "in_out"

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v not in OPTIONS[f.name]:
                raise ValueError(f"ME convention {f.name}={v!r}: expected one of {OPTIONS[f.name]}")

    @property
    def is_default(self) -> bool:
        return self == MEConventions()

    def describe(self) -> str:
        return ",".join(f"{k}={v}" for k, v in asdict(self).items())


# The one line to change when a real checkpoint says otherwise.
DEFAULT = MEConventions()


def parse(spec) -> MEConventions:
    """None / "" -> DEFAULT; an MEConventions -> itself; a dict or "key=value,key=value" -> DEFAULT with those fields
    replaced (cfg["MODEL"]["ME_CONVENTIONS"], ``predict.py --me-conventions``)."""
    if spec is None or spec == "":
        return DEFAULT
    if isinstance(spec, MEConventions):
        return spec
    if isinstance(spec, str):
        items = {}
        for part in spec.split(","):
            if not part.strip():
                continue
            if "=" not in part:
                raise ValueError(f"ME convention {part!r}: expected key=value")
            k, v = part.split("=", 1)
            items[k.strip()] = v.strip()
        spec = items
    unknown = set(spec) - set(OPTIONS)
    if unknown:
        raise ValueError(f"unknown ME convention(s) {sorted(unknown)}: expected {sorted(OPTIONS)}")
    return replace(DEFAULT, **spec)


def all_combinations():
    """Every combination of the options (32), the default first."""
    keys = list(OPTIONS)
    out = [MEConventions(**dict(zip(keys, vals))) for vals in itertools.product(*(OPTIONS[k] for k in keys))]
    out.sort(key=lambda c: (not c.is_default,))
    return out


# ------------------------------------------------------------------------------------------------------------------
# kernel index  <->  geometric offset
# ------------------------------------------------------------------------------------------------------------------
KSIZE = {"conv5": (5, 5, 5, 1), "conv3": (3, 3, 3, 3), "down": (2, 2, 2, 1), "up": (2, 2, 2, 1)}


def layer_kind(name: str) -> str:
    """Kind of the convolution whose state_dict name (without ".kernel") is ``name`` (minkunet.py:55-159)."""
    if name == "conv0p1s1":
        return "conv5"
    if name.startswith("convtr"):
        return "up"
    if name == "final" or ".downsample." in name:
        return "lin"
    if name.startswith("conv") and name[4].isdigit():
        return "down"
    if name.endswith(".conv1") or name.endswith(".conv2"):
        return "conv3"
    raise ValueError(f"not a convolution of CustomMinkUNet: {name!r}")


def index_offsets(kind: str, cv: MEConventions) -> np.ndarray:
    """[K, 4] integer (dx, dy, dz, dt) in units of the layer's input stride: the geometric offset ``in - out``
    (for a transposed convolution: ``fine - coarse`` of the stride map it shares) that the checkpoint's weight slice
    ``W[k]`` is applied to under convention ``cv``."""
    ks = KSIZE[kind]
    axes = []
    for k in ks:
        if k % 2 == 1:
            a = [i - k // 2 for i in range(k)]
            if cv.odd_kernel_sign == "minus":
                a = [-v for v in a]
        else:
            a = list(range(k))
            if cv.even_kernel_order == "descending":
                a = a[::-1]
        axes.append(a)
    if cv.offset_order == "x_fastest":
        offs = [(x, y, z, t) for t in axes[3] for z in axes[2] for y in axes[1] for x in axes[0]]
    else:
        offs = [(x, y, z, t) for x in axes[0] for y in axes[1] for z in axes[2] for t in axes[3]]
    offs = np.asarray(offs, dtype=np.int64)
    if kind == "up" and cv.transpose_index == "mirrored":
        offs = offs[::-1].copy()
    return offs


def kernel_index_map(kind: str, cv: MEConventions) -> np.ndarray:
    """kperm [K]: the internal (canonical) kernel index k uses the checkpoint's slice ``W_ckpt[kperm[k]]``."""
    canon = index_offsets(kind, MEConventions())
    theirs = index_offsets(kind, cv)
    where = {tuple(o): j for j, o in enumerate(theirs.tolist())}
    assert len(where) == len(theirs) == len(canon), "a convention must enumerate every offset exactly once"
    return np.asarray([where[tuple(o)] for o in canon.tolist()], dtype=np.int64)


def blob_permutation(layout, shapes: dict, cv: MEConventions) -> np.ndarray | None:
    """perm (int64 [numel]) with ``blob_internal = blob_ckpt[perm]`` for the flat weight blob described by ``layout``
    ((name, offset, numel), ... from ``_native.weight_layout``) and the kernel shapes ``shapes[name] = (K, C_in, C_out)``;
    None when ``cv`` is the canonical convention (nothing to move)."""
    if cv.is_default:
        return None
    total = max(off + num for _, off, num in layout)
    perm = np.arange(total, dtype=np.int64)
    for name, off, num in layout:
        if not name.endswith(".kernel"):
            continue
        conv = name[: -len(".kernel")]
        kind = layer_kind(conv)
        K, cin, cout = shapes[name]
        assert K * cin * cout == num, (name, K, cin, cout, num)
        if kind == "lin":
            if cv.lin_layout == "out_in":
                # internal [ci][co] <- checkpoint [co][ci]
                ci, co = np.meshgrid(np.arange(cin), np.arange(cout), indexing="ij")
                perm[off: off + num] = off + (co * cin + ci).reshape(-1)
            continue
        kperm = kernel_index_map(kind, cv)
        assert len(kperm) == K, (name, K, len(kperm))
        perm[off: off + num] = off + (kperm[:, None] * (cin * cout) + np.arange(cin * cout)[None, :]).reshape(-1)
    return perm


def inverse_permutation(perm: np.ndarray) -> np.ndarray:
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm), dtype=perm.dtype)
    return inv
