"""ctypes wrapper of oracle/libsps_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by sps_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsps_oracle.so")


def _load():
    src = os.path.join(HERE, "sps_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    lib = C.CDLL(LIB)
    lib.sps_oracle_numel.restype = C.c_int64
    lib.sps_oracle_tensor_info.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.sps_oracle_forward.restype = C.c_int64
    lib.sps_oracle_forward.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


lib = _load()


def layout():
    out = []
    buf = C.create_string_buffer(128)
    off, num = C.c_int64(), C.c_int64()
    for i in range(lib.sps_oracle_num_tensors()):
        lib.sps_oracle_tensor_info(i, buf, 128, C.byref(off), C.byref(num))
        out.append((buf.value.decode(), off.value, num.value))
    return out


def pack_blob(params: dict) -> np.ndarray:
    """oracle parameter dict (reference state_dict names) -> flat float32 blob."""
    blob = np.empty(lib.sps_oracle_numel(), dtype=np.float32)
    for name, off, numel in layout():
        v = np.ascontiguousarray(params[name], dtype=np.float32).reshape(-1)
        assert v.size == numel, (name, v.size, numel)
        blob[off: off + numel] = v
    return blob


def max_threads() -> int:
    return lib.sps_oracle_max_threads()


def forward(blob: np.ndarray, coords: np.ndarray, voxel_size: float, nthreads: int = 0, want_details: bool = True):
    """Returns (scores [n], info dict with voxels/inverse/logits/level_counts/timings)."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    n, ld = coords.shape
    scores = np.empty(n, dtype=np.float32)
    vox = np.empty((max(n, 1), 5), dtype=np.int32) if want_details else None
    inv = np.empty(max(n, 1), dtype=np.int64) if want_details else None
    logits = np.empty(max(n, 1), dtype=np.float32) if want_details else None
    counts = np.zeros(5, dtype=np.int64)
    timings = np.zeros(4, dtype=np.float64)

    def ptr(a):
        return a.ctypes.data if a is not None else None

    v = lib.sps_oracle_forward(ptr(coords), n, ld, voxel_size, ptr(blob), ptr(scores), nthreads,
                               ptr(vox), ptr(inv), ptr(logits), ptr(counts), ptr(timings))
    if v < 0:
        raise RuntimeError(f"sps_oracle_forward failed: {v}")
    info = dict(level_counts=counts.tolist(), timings=timings.tolist())
    if want_details:
        info.update(voxels=vox[:v], inverse=inv[:n], logits=logits[:v])
    return scores, info
