"""ctypes binding of libsps_hip.so (C ABI: include/sps_hip.h).

The product path has NO fallback: if the library is missing or fails to load, importing
this module raises.  Device pointers are passed as integers (``tensor.data_ptr()``)."""
from __future__ import annotations

import ctypes as C
import os

from . import _build

NUM_LEVELS = 5
SPS_OK = 0
ERR_RANGE = -4
ERR_NOMEM = -3
ERR_INVALID = -1
ERR_ITEMCAP = -6


class SpsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsps_hip error {code}: {msg}")
        self.code = code


ABI_VERSION = 201          # sps_version() of the library this binding was written against


def _load() -> C.CDLL:
    # SPS_LIB: diagnostic builds (tools/ablate*.sh, tune_conv.sh ...) live at their own path and never overwrite
    # the product library
    path = os.environ.get("SPS_LIB") or _build.LIB
    if path == _build.LIB and (not os.path.exists(path) or _build.is_stale()):
        # an edited .hip / .inc.h must never run against an old binary: rebuild when hipcc is here (it cross-compiles
        # without a GPU); a box without hipcc runs the shipped .so, whose ABI version is checked below.
        # Never falls back to CPU code.
        if _build.have_hipcc():
            path = _build.build()
        elif not os.path.exists(path):
            raise ImportError("libsps_hip.so is missing and hipcc is not available to build it")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 / libhsa-runtime64.  Loaded AFTER torch, this library binds
    # to the copies torch already mapped (same SONAME); loaded BEFORE it, /opt/rocm's copies come in first and the process ends
    # up with two runtimes, of which the second finds "no ROCm-capable device" (seen with __graft_entry__.build() followed by
    # smoke() in one process).  The host side of this package is PyTorch's anyway (device memory, streams).
    import importlib.util
    if importlib.util.find_spec("torch") is not None:     # (the ctypes binding alone works without torch)
        import torch  # noqa: F401
    lib = C.CDLL(path)
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    sig = {
        "sps_last_error": (C.c_char_p, []),
        "sps_version": (i32, []),
        "sps_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "sps_ctx_destroy": (i32, [vp]),
        "sps_reserve": (i32, [vp, i64]),
        "sps_ctx_set_level_fractions": (i32, [vp, vp]),
        "sps_ctx_set_inference_only": (i32, [vp, i32]),
        "sps_ctx_set_pipelined": (i32, [vp, i32]),
        "sps_arena_bytes": (i64, [vp]),
        "sps_weights_num_tensors": (i32, []),
        "sps_weights_tensor_info": (i32, [i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64)]),
        "sps_weights_numel": (i64, []),
        "sps_weights_load": (i32, [vp, vp, i64]),
        "sps_weights_create": (i32, [i32, vp, i64, i32, C.POINTER(vp)]),
        "sps_weights_destroy": (i32, [vp]),
        "sps_ctx_set_weights": (i32, [vp, vp]),
        "sps_forward": (i32, [vp, vp, i64, i64, f32, vp, vp]),
        "sps_forward_metrics": (i32, [vp, vp, i64, i64, f32, f32, i32, vp, vp, vp]),
        "sps_head_num_tensors": (i32, [i32]),
        "sps_head_tensor_info": (i32, [i32, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64)]),
        "sps_head_numel": (i64, [i32]),
        "sps_weights_load_head": (i32, [vp, vp, i64, i32]),
        "sps_forward_head": (i32, [vp, vp, i64, i64, f32, vp, f32, vp, i64, i32, vp]),
        "sps_check": (i32, [vp, vp]),
        "sps_metrics": (i32, [vp, vp, vp, i64, i64, f32, i32, C.POINTER(C.c_double), vp]),
        "sps_metrics_dev": (i32, [vp, vp, vp, i64, i64, f32, i32, vp, vp]),
        "sps_profile_enable": (i32, [vp, i32]),
        "sps_profile_count": (i32, [vp]),
        "sps_profile_read": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(f32)]),
        "sps_profile_kernel": (i32, [vp, i32, C.c_char_p, i32]),
        "sps_map_upload": (i32, [vp, vp, i64, i64, f32, vp]),
        "sps_map_upload_voxels": (i32, [vp, vp, i64, i64, vp]),
        "sps_submap_voxel": (i32, [vp, vp, i64, i64, vp, C.POINTER(i64), C.POINTER(i64), vp]),
        "sps_submap_voxel_ijk": (i32, [vp, vp, i64, i64, f32, vp, C.POINTER(i64), C.POINTER(i64), vp]),
        "sps_transform_points": (i32, [vp, vp, i32, i64, i64, vp, vp, i32, i64, vp]),
        "sps_filter_prepare": (i32, [vp, vp, i32, i64, i64, vp, vp, vp, vp]),
        "sps_forward_n": (i32, [vp, vp, i64, i64, vp, f32, vp, vp]),
        "sps_compact_stable": (i32, [vp, vp, vp, i64, i32, i64, f32, vp, vp, vp]),
        "sps_train_forward": (i32, [vp, vp, i64, vp, i64, i64, f32, vp, vp, vp]),
        "sps_train_backward": (i32, [vp, vp, vp, vp, i64, vp]),
        "sps_train_generation": (i32, [vp, C.POINTER(i64)]),
        "sps_train_backward_at": (i32, [vp, i64, vp, vp, vp, i64, vp]),
        "sps_scan_mse": (i32, [vp, vp, i64, vp, i64, i64, vp, vp, vp]),
        "sps_scan_mse_backward": (i32, [vp, vp, i64, vp, i64, i64, vp, vp, vp, vp]),
        "sps_radius_grid_upload": (i32, [vp, vp, vp, vp, vp, i64, i64, C.c_double, C.c_double, vp]),
        "sps_radius_count": (i32, [vp, vp, i64, i64, vp, vp]),
        "sps_radius_fill": (i32, [vp, vp, i64, i64, vp, vp, vp]),
        "sps_radius_grid_attach": (i32, [vp, vp]),
        "sps_radius_item": (i32, [vp, vp, i32, i64, i64, f32, vp, vp, i64, i64, vp, vp]),
        "sps_forward_metrics_n": (i32, [vp, vp, i64, i64, vp, f32, f32, i32, vp, vp, vp]),
        "sps_level_counts": (i32, [vp, C.POINTER(i64)]),
        "sps_get_voxels": (i32, [vp, i32, vp]),
        "sps_get_inverse": (i32, [vp, vp]),
        "sps_get_parent": (i32, [vp, i32, vp]),
        "sps_get_map_pairs": (i32, [vp, i32, C.POINTER(i64)]),
        "sps_get_tile_masks": (i32, [vp, i32, vp, C.POINTER(i64)]),
        "sps_get_nbr": (i32, [vp, i32, vp]),
        "sps_get_kernel_map": (i32, [vp, i32, i32, vp, C.POINTER(i64)]),
        "sps_get_logits": (i32, [vp, vp]),
        "sps_get_feature": (i32, [vp, C.c_char_p, vp, C.POINTER(i64), C.POINTER(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.sps_version() != ABI_VERSION:
        raise ImportError(f"{path} reports ABI version {lib.sps_version()}, this binding needs {ABI_VERSION}: rebuild it "
                          "(python -m sps_amd._build)")
    return lib


lib = _load()
EXPORTS = ["sps_last_error", "sps_version", "sps_ctx_create", "sps_ctx_destroy", "sps_reserve",
           "sps_ctx_set_level_fractions", "sps_ctx_set_inference_only", "sps_ctx_set_pipelined", "sps_arena_bytes",
           "sps_weights_num_tensors", "sps_weights_tensor_info", "sps_weights_numel", "sps_weights_load",
           "sps_weights_create", "sps_weights_destroy", "sps_ctx_set_weights",
           "sps_forward", "sps_forward_metrics", "sps_head_num_tensors", "sps_head_tensor_info", "sps_head_numel", "sps_weights_load_head",
           "sps_forward_head", "sps_check", "sps_metrics", "sps_metrics_dev",
           "sps_profile_enable", "sps_profile_count", "sps_profile_read", "sps_profile_kernel", "sps_map_upload", "sps_map_upload_voxels",
           "sps_submap_voxel", "sps_submap_voxel_ijk", "sps_transform_points", "sps_filter_prepare", "sps_forward_n",
           "sps_compact_stable", "sps_train_forward", "sps_train_backward", "sps_train_generation", "sps_train_backward_at", "sps_scan_mse", "sps_scan_mse_backward", "sps_radius_grid_upload", "sps_radius_count",
           "sps_radius_fill", "sps_radius_grid_attach", "sps_radius_item", "sps_forward_metrics_n", "sps_level_counts", "sps_get_voxels",
           "sps_get_inverse", "sps_get_parent", "sps_get_map_pairs", "sps_get_tile_masks", "sps_get_nbr", "sps_get_kernel_map", "sps_get_logits", "sps_get_feature"]


def check(rc: int) -> None:
    if rc != SPS_OK:
        raise SpsError(rc, lib.sps_last_error().decode())


_LAYOUTS: dict = {}


def weight_layout(out_channels: int = 1):
    """((name, offset, numel), ...) of the weight blob, from the library itself (asked once per head width)."""
    cached = _LAYOUTS.get(out_channels)
    if cached is not None:
        return cached
    out = []
    buf = C.create_string_buffer(128)
    off, num = C.c_int64(), C.c_int64()
    n = lib.sps_head_num_tensors(out_channels)
    if n < 0:
        check(n)
    for i in range(n):
        check(lib.sps_head_tensor_info(out_channels, i, buf, 128, C.byref(off), C.byref(num)))
        out.append((buf.value.decode(), off.value, num.value))
    _LAYOUTS[out_channels] = tuple(out)
    return _LAYOUTS[out_channels]


class Weights:
    """One device-resident weight set (sps_weights_create): uploaded once per device and parameter version, attached
    to any number of contexts of that device in O(1)."""

    def __init__(self, device: int, blob_host_ptr: int, numel: int, out_channels: int = 1):
        h = C.c_void_p()
        check(lib.sps_weights_create(int(device), blob_host_ptr, int(numel), int(out_channels), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            lib.sps_weights_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Owns one ``sps_ctx`` (one per process, device and stream)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        check(lib.sps_ctx_create(device, C.byref(h)))
        self.handle = h
        self.device = device
        self.weights = None

    def close(self):
        if getattr(self, "handle", None):
            lib.sps_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- thin wrappers ----------------------------------------------------------------
    def reserve(self, max_points: int):
        check(lib.sps_reserve(self.handle, int(max_points)))

    LIDAR_FRACTIONS = (1.0, 0.6, 0.3, 0.15, 0.08)      # V_l / V_0 of 0.1 m LiDAR clouds is ~0.39 / 0.14 / 0.05 / 0.015

    def set_level_fractions(self, frac=None):
        """Compact arena (include/sps_hip.h): level l holds frac[l] * max_points rows; None = full size (never overflows)."""
        arr = None if frac is None else (C.c_float * NUM_LEVELS)(*[float(x) for x in frac])
        check(lib.sps_ctx_set_level_fractions(self.handle, arr))

    def set_inference_only(self, on: bool = True):
        """Inference-only context (include/sps_hip.h): at the pair-exact levels the rulebook replaces the neighbour table
        (less arena, fewer bytes written per scan); training on the context switches it back."""
        check(lib.sps_ctx_set_inference_only(self.handle, 1 if on else 0))

    def set_pipelined(self, on: bool = True):
        """Launch-geometry hint (include/sps_hip.h): True = forwards of several contexts run beside each other (least work per
        forward), False (default) = one forward after another (shortest chain).  Results are bit-identical either way."""
        check(lib.sps_ctx_set_pipelined(self.handle, 1 if on else 0))

    def arena_bytes(self) -> int:
        return int(lib.sps_arena_bytes(self.handle))

    def load_weights(self, blob_host_ptr: int, numel: int, out_channels: int = 1):
        check(lib.sps_weights_load_head(self.handle, blob_host_ptr, int(numel), int(out_channels)))
        self.weights = None

    def set_weights(self, weights: "Weights"):
        check(lib.sps_ctx_set_weights(self.handle, weights.handle))
        self.weights = weights          # keeps the Python owner alive as long as the context uses it

    def forward_head(self, coords_ptr: int, ld: int, n: int, voxel_size: float, feats_ptr, t_base: float,
                     out_ptr: int, ldo: int, activation: int, stream: int):
        check(lib.sps_forward_head(self.handle, coords_ptr, ld, n, voxel_size, feats_ptr, t_base, out_ptr, ldo,
                                   activation, stream))

    def forward_metrics(self, batch_ptr: int, ld: int, n: int, voxel_size: float, eps: float, n_batches: int,
                        scores_ptr: int, out_ptr: int, stream: int):
        check(lib.sps_forward_metrics(self.handle, batch_ptr, ld, n, voxel_size, eps, n_batches, scores_ptr, out_ptr, stream))

    def forward_metrics_n(self, batch_ptr: int, ld: int, n_max: int, n_dev_ptr: int, voxel_size: float, eps: float,
                          n_batches: int, scores_ptr: int, out_ptr: int, stream: int):
        check(lib.sps_forward_metrics_n(self.handle, batch_ptr, ld, n_max, n_dev_ptr, voxel_size, eps, n_batches, scores_ptr,
                                        out_ptr, stream))

    def radius_grid_attach(self, owner: "Context"):
        check(lib.sps_radius_grid_attach(self.handle, owner.handle))
        self._grid_owner = owner        # the owner keeps the device copies: it must outlive this view

    def radius_item(self, scan_ptr: int, in_f64: bool, ld: int, n: int, batch_index: float, row_off_ptr, rows_ptr: int,
                    ldo: int, row_cap: int, n_rows_ptr: int, stream: int):
        check(lib.sps_radius_item(self.handle, scan_ptr, int(in_f64), ld, n, float(batch_index), row_off_ptr, rows_ptr, ldo,
                                  row_cap, n_rows_ptr, stream))

    def forward(self, coords_ptr: int, ld: int, n: int, voxel_size: float, scores_ptr: int, stream: int):
        check(lib.sps_forward(self.handle, coords_ptr, ld, n, voxel_size, scores_ptr, stream))

    def check_errors(self, stream: int):
        check(lib.sps_check(self.handle, stream))

    def metrics(self, scores_ptr: int, batch_ptr: int, ld: int, n: int, eps: float, n_batches: int, stream: int):
        out = (C.c_double * (8 * n_batches))()
        check(lib.sps_metrics(self.handle, scores_ptr, batch_ptr, ld, n, eps, n_batches, out, stream))
        return [list(out[8 * b: 8 * b + 8]) for b in range(n_batches)]

    def metrics_dev(self, scores_ptr: int, batch_ptr: int, ld: int, n: int, eps: float, n_batches: int,
                    out_ptr: int, stream: int):
        check(lib.sps_metrics_dev(self.handle, scores_ptr, batch_ptr, ld, n, eps, n_batches, out_ptr, stream))

    def profile_enable(self, on: bool):
        check(lib.sps_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        """[(stage name, milliseconds)] of the last forward (synchronises on the stage events)."""
        out = []
        buf = C.create_string_buffer(96)
        ms = C.c_float()
        for i in range(lib.sps_profile_count(self.handle)):
            check(lib.sps_profile_read(self.handle, i, buf, 96, C.byref(ms)))
            out.append((buf.value.decode(), ms.value))
        return out

    def profile_kernels(self):
        """[kernel class] of the stages profile_read() lists (sps_profile_kernel)."""
        buf = C.create_string_buffer(128)
        out = []
        for i in range(lib.sps_profile_count(self.handle)):
            check(lib.sps_profile_kernel(self.handle, i, buf, 128))
            out.append(buf.value.decode())
        return out

    def map_upload(self, xyz_ptr: int, ld: int, m: int, ds: float, stream: int):
        check(lib.sps_map_upload(self.handle, xyz_ptr, ld, m, ds, stream))

    def map_upload_voxels(self, ijk_ptr: int, ld: int, m: int, stream: int):
        check(lib.sps_map_upload_voxels(self.handle, ijk_ptr, ld, m, stream))

    def submap_voxel(self, scan_ptr: int, ld: int, n: int, out_ptr: int, stream: int):
        a, b = C.c_int64(), C.c_int64()
        check(lib.sps_submap_voxel(self.handle, scan_ptr, ld, n, out_ptr, C.byref(a), C.byref(b), stream))
        return a.value, b.value

    def submap_voxel_ijk(self, scan_ptr: int, ld: int, n: int, ds: float, out_ptr: int, stream: int):
        a, b = C.c_int64(), C.c_int64()
        check(lib.sps_submap_voxel_ijk(self.handle, scan_ptr, ld, n, ds, out_ptr, C.byref(a), C.byref(b), stream))
        return a.value, b.value

    @staticmethod
    def _mat(T):
        """4x4 host matrix -> ctypes double[16] (row-major), None = identity."""
        if T is None:
            return None
        import numpy as np
        a = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(16))
        return (C.c_double * 16)(*a.tolist())

    def transform_points(self, xyz_ptr: int, in_f64: bool, ld: int, n: int, T, out_ptr: int, out_f64: bool, ldo: int,
                         stream: int):
        check(lib.sps_transform_points(self.handle, xyz_ptr, int(in_f64), ld, n, self._mat(T), out_ptr, int(out_f64), ldo,
                                       stream))

    def filter_prepare(self, raw_ptr: int, in_f64: bool, ld: int, n: int, T, batch_ptr: int, counts_ptr: int, stream: int):
        check(lib.sps_filter_prepare(self.handle, raw_ptr, int(in_f64), ld, n, self._mat(T), batch_ptr, counts_ptr, stream))

    def forward_n(self, coords_ptr: int, ld: int, n_max: int, n_dev_ptr: int, voxel_size: float, scores_ptr: int, stream: int):
        check(lib.sps_forward_n(self.handle, coords_ptr, ld, n_max, n_dev_ptr, voxel_size, scores_ptr, stream))

    def compact_stable(self, scores_ptr: int, rows_ptr: int, ld: int, cols: int, n: int, eps: float, out_ptr: int,
                       count_ptr: int, stream: int):
        check(lib.sps_compact_stable(self.handle, scores_ptr, rows_ptr, ld, cols, n, eps, out_ptr, count_ptr, stream))

    def train_forward(self, params_ptr: int, numel: int, coords_ptr: int, ld: int, n: int, voxel_size: float,
                      scores_ptr: int, batch_stats_ptr, stream: int):
        check(lib.sps_train_forward(self.handle, params_ptr, numel, coords_ptr, ld, n, voxel_size, scores_ptr,
                                    batch_stats_ptr, stream))

    def train_backward(self, dscores_ptr: int, scores_ptr: int, grad_ptr: int, numel: int, stream: int, generation=None):
        """``generation`` (train_generation() right after the forward): refuse to differentiate a forward whose activations
        a later forward on this context has overwritten."""
        if generation is None:
            check(lib.sps_train_backward(self.handle, dscores_ptr, scores_ptr, grad_ptr, numel, stream))
        else:
            check(lib.sps_train_backward_at(self.handle, int(generation), dscores_ptr, scores_ptr, grad_ptr, numel, stream))

    def train_generation(self) -> int:
        g = C.c_int64()
        check(lib.sps_train_generation(self.handle, C.byref(g)))
        return g.value

    def level_counts(self):
        out = (C.c_int64 * NUM_LEVELS)()
        check(lib.sps_level_counts(self.handle, out))
        return list(out)

    def kernel_map(self, which: int, source: int = 0):
        """Dense int32 [K, V_out] table of a kernel map of the last forward (include/sps_hip.h: sps_get_kernel_map), on the
        device; with source = 1 also the number of pairs the rulebook holds."""
        import torch
        counts = self.level_counts()
        level = which if which <= 4 else (0 if which == 5 else which - 5)
        K = 81 if which <= 4 else (125 if which == 5 else 8)
        out = torch.empty((K, counts[level]), dtype=torch.int32, device=f"cuda:{self.device}")
        n = C.c_int64()
        check(lib.sps_get_kernel_map(self.handle, which, source, out.data_ptr(), C.byref(n)))
        return (out, n.value) if source == 1 else out

    def map_pairs(self, which: int):
        out = (C.c_int64 * 125)()
        check(lib.sps_get_map_pairs(self.handle, which, out))
        return list(out)[: 125 if which == 5 else 81]
