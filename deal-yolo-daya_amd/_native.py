"""ctypes binding of libdyd_gfx950.so (C ABI: include/dyd.h).

This is the thin layer between the Python host code and the HIP kernels.  It has NO CPU
fallback: a missing library or a missing gfx950 device raises ``NativeUnavailable`` the
first time a device stage is requested.

Numpy-facing helpers (``bbox_minmax`` ...) use the host-pointer entry points (the library
stages H2D/D2H); ``lib()`` exposes the raw ``_dev`` entry points for callers that keep data
resident in HBM (bench.py, the distributed path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DYD_LIB_PATH") or os.path.join(_PKG, "libdyd_gfx950.so")   # env: A/B builds of the library

KEEP_FIRST, KEEP_LAST, KEEP_NONE = 0, 1, 2
_KEEP = {"first": KEEP_FIRST, "last": KEEP_LAST, False: KEEP_NONE}

_lock = threading.Lock()
_lib = None
_ready = False


class NativeUnavailable(RuntimeError):
    """libdyd_gfx950.so could not be loaded or no gfx950 device is usable."""


class NativeError(RuntimeError):
    """An entry point of libdyd_gfx950.so returned an error code."""


_c = C
_dp, _i32p, _i64p, _u8p, _u64p = (C.POINTER(t) for t in (C.c_double, C.c_int32, C.c_int64, C.c_uint8, C.c_uint64))

# name -> (restype, argtypes); mirrors include/dyd.h line by line
SIGNATURES = {
    "dyd_init": (C.c_int, [C.c_int]),
    "dyd_shutdown": (None, []),
    "dyd_last_error": (C.c_char_p, []),
    "dyd_device_count": (C.c_int, []),
    "dyd_version": (C.c_char_p, []),
    "dyd_device_name": (C.c_char_p, []),
    "dyd_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "dyd_free": (C.c_int, [C.c_void_p]),
    "dyd_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dyd_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dyd_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "dyd_sync": (C.c_int, [C.c_void_p]),
    "dyd_device_status": (C.c_int, [C.c_void_p]),
    "dyd_last_kernel_ms": (C.c_double, []),
    "dyd_bbox_minmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dyd_bbox_minmax_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_iou_any_ge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]),
    "dyd_iou_any_ge_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "dyd_bbox_iou_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "dyd_host_pool_trim": (None, []),
    "dyd_stage_acquire": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "dyd_stage_release": (None, [C.c_void_p]),
    "dyd_bbox_iou_fused_staged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_bbox_iou_fused_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                         C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_hash128": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dyd_hash128_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dyd_dedup": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dyd_dedup_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "dyd_isin": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "dyd_dedup_partner": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "dyd_isin_partner": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "dyd_host_cells_differ": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dyd_isin_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dyd_dedup_global_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "dyd_mt19937_permutation": (C.c_int, [C.c_uint32, C.c_int64, C.c_void_p]),
    "dyd_split_ids": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                C.c_void_p, C.c_void_p]),
    "dyd_split_ids_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_split_ids_sharded_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_mt19937_permutation_dev": (C.c_int, [C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_split_ids_seeded": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_void_p]),
    "dyd_split_ids_seeded_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_yolo_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "dyd_yolo_lines_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "dyd_json_scan_polygons": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "dyd_json_emit_polygons": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dyd_json_scan_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "dyd_scan_n_boxes": (C.c_int64, [C.c_void_p]),
    "dyd_scan_n_points": (C.c_int64, [C.c_void_p]),
    "dyd_scan_xy": (C.c_void_p, [C.c_void_p]),
    "dyd_scan_pt_off": (C.c_void_p, [C.c_void_p]),
    "dyd_scan_cell_box_off": (C.c_void_p, [C.c_void_p]),
    "dyd_scan_status": (C.c_void_p, [C.c_void_p]),
    "dyd_scan_wh_kind": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dyd_scan_wh_value": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dyd_scan_iou_host": (C.c_void_p, [C.c_void_p]),
    "dyd_scan_fast_cells": (C.c_int64, [C.c_void_p]),
    "dyd_json_scan_polygons_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "dyd_json_replace_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double,
                                       C.c_int, C.POINTER(C.c_void_p)]),
    "dyd_scan_high": (C.c_void_p, [C.c_void_p]),
    "dyd_scan_parts": (C.c_int32, [C.c_void_p]),
    "dyd_scan_part": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p)]),
    "dyd_scan_text": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dyd_scan_totals": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dyd_scan_free": (None, [C.c_void_p]),
    "dyd_synth_json": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                 C.POINTER(C.c_void_p), C.c_void_p]),
    "dyd_json_scan_labelled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                         C.POINTER(C.c_void_p)]),
    "dyd_scan_sel": (C.c_void_p, [C.c_void_p]),
    "dyd_json_split_expand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "dyd_split_status": (C.c_void_p, [C.c_void_p]),
    "dyd_split_n_expanded": (C.c_void_p, [C.c_void_p]),
    "dyd_split_rows": (C.c_int64, [C.c_void_p]),
    "dyd_split_row_cell": (C.c_void_p, [C.c_void_p]),
    "dyd_split_row_label": (C.c_void_p, [C.c_void_p]),
    "dyd_split_events": (C.c_int64, [C.c_void_p]),
    "dyd_split_event_cell": (C.c_void_p, [C.c_void_p]),
    "dyd_split_event_kind": (C.c_void_p, [C.c_void_p]),
    "dyd_split_strings": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dyd_split_free": (None, [C.c_void_p]),
    "dyd_json_split_expand_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int,
                                          C.POINTER(C.c_void_p)]),
    "dyd_split_rec_views": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dyd_split_event_code": (C.c_void_p, [C.c_void_p]),
    "dyd_split_undefined": (C.c_int64, [C.c_void_p]),
    "dyd_split_label_first": (C.c_void_p, [C.c_void_p]),
    "dyd_split_label_count": (C.c_void_p, [C.c_void_p]),
    "dyd_split_fast_cells": (C.c_int64, [C.c_void_p]),
    "dyd_split_all_ascii": (C.c_int, [C.c_void_p]),
    "dyd_split_reason_code": (C.c_void_p, [C.c_void_p]),
    "dyd_split_reason_distinct": (C.c_int64, [C.c_void_p]),
    "dyd_split_reason_first": (C.c_void_p, [C.c_void_p]),
    "dyd_split_seconds": (None, [C.c_void_p, C.c_void_p]),
    "dyd_json_relabel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_int, C.POINTER(C.c_void_p)]),
    "dyd_relabel_status": (C.c_void_p, [C.c_void_p]),
    "dyd_relabel_has_diff": (C.c_void_p, [C.c_void_p]),
    "dyd_relabel_counts": (C.c_void_p, [C.c_void_p]),
    "dyd_relabel_tokens": (C.c_int64, [C.c_void_p]),
    "dyd_relabel_token_cell": (C.c_void_p, [C.c_void_p]),
    "dyd_relabel_strings": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dyd_relabel_free": (None, [C.c_void_p]),
    "dyd_csv_index": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "dyd_csv_rows": (C.c_int64, [C.c_void_p]),
    "dyd_csv_cols": (C.c_int32, [C.c_void_p]),
    "dyd_csv_header": (C.c_int64, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "dyd_csv_extract": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dyd_csv_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "dyd_csv_col_bytes": (C.c_int64, [C.c_void_p, C.c_int32]),
    "dyd_csv_row_end": (C.c_int64, [C.c_void_p, C.c_int64]),
    "dyd_csv_has_cr": (C.c_int, [C.c_void_p]),
    "dyd_csv_free": (None, [C.c_void_p]),
    "dyd_csv_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "dyd_host_free": (None, [C.c_void_p]),
    "dyd_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "dyd_membench_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
}


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libdyd_gfx950.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-j8"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libdyd_gfx950.so failed:\n" + out.stdout)
    if verbose:
        print(out.stdout)
    return LIB_PATH


def _one_hip_runtime_per_process():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If this library pulled in
    /opt/rocm's copy first and torch initialised its bundled copy afterwards, the process would hold
    two HIP runtimes and the second one finds no GPU.  Importing torch first (when it is installed)
    makes libdyd_gfx950.so bind to the runtime torch uses, so device pointers and streams are shared."""
    import sys

    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing; without it /opt/rocm's runtime is used
            pass


def load_library():
    """dlopen the library and declare every prototype.  Does not touch the GPU."""
    global _lib
    with _lock:
        if _lib is None:
            _one_hip_runtime_per_process()
            if not os.path.exists(LIB_PATH):
                raise NativeUnavailable(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C deal-yolo-daya_amd/csrc`).  There is no CPU fallback.")
            try:
                lib_ = C.CDLL(LIB_PATH)
            except OSError as e:
                raise NativeUnavailable(f"cannot load {LIB_PATH}: {e}") from e
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib_, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib_
    return _lib


def lib():
    """The loaded library with an initialised device context (raises if there is no gfx950)."""
    global _ready
    l = load_library()
    if not _ready:
        with _lock:
            if not _ready:
                dev = int(os.environ.get("DYD_DEVICE", "-1"))
                rc = l.dyd_init(dev)
                if rc != 0:
                    raise NativeUnavailable(
                        "dyd_init failed: " + l.dyd_last_error().decode("utf-8", "replace")
                        + " — the HIP device stage is required; there is no CPU fallback.")
                _ready = True
    return l


def available() -> bool:
    try:
        lib()
        return True
    except NativeUnavailable:
        return False


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what} failed ({rc}): " + load_library().dyd_last_error().decode("utf-8", "replace"))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_name() -> str:
    return lib().dyd_device_name().decode()


def last_kernel_ms() -> float:
    return float(load_library().dyd_last_kernel_ms())


# ------------------------------------------------------------------ numpy-facing helpers
def bbox_minmax(xy: np.ndarray, pt_off: np.ndarray):
    """K1 over host arrays: returns (box4 [B,4] f64, arg4 [B,4] i32)."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    pt_off = np.ascontiguousarray(pt_off, dtype=np.int32)
    nb = len(pt_off) - 1
    if nb < 0 or (nb >= 0 and (pt_off[0] != 0 or 2 * int(pt_off[-1]) != xy.size)):
        raise ValueError("pt_off must start at 0 and end at the number of points")
    box = np.empty((nb, 4), np.float64)
    arg = np.empty((nb, 4), np.int32)
    check(lib().dyd_bbox_minmax(_ptr(xy), _ptr(pt_off), nb, _ptr(box), _ptr(arg)), "dyd_bbox_minmax")
    return box, arg


def iou_any_ge(box4: np.ndarray, row_off: np.ndarray, min_boxes: int, thr: float, want_max: bool = False):
    """K2 over host arrays: HIGH flag per row (and the max pair IoU when ``want_max``)."""
    box4 = np.ascontiguousarray(box4, dtype=np.float64).reshape(-1)
    row_off = np.ascontiguousarray(row_off, dtype=np.int32)
    n = len(row_off) - 1
    if n < 0 or row_off[0] != 0 or 4 * int(row_off[-1]) != box4.size:
        raise ValueError("row_off must start at 0 and end at the number of boxes")
    high = np.empty(n, np.uint8)
    mx = np.empty(n, np.float64) if want_max else None
    check(lib().dyd_iou_any_ge(_ptr(box4), _ptr(row_off), n, int(min_boxes), float(thr), _ptr(high),
                               _ptr(mx) if want_max else None), "dyd_iou_any_ge")
    return (high, mx) if want_max else high


def bbox_iou_fused(xy: np.ndarray, pt_off: np.ndarray, box_off: np.ndarray, min_boxes: int, thr: float,
                   want_box: bool = False):
    """Fused K1+K2 over host arrays (one launch): (arg4 [B,4] i32, high [N] u8[, box4 [B,4] f64]).  The flag follows the
    reference's replace -> IoU chain: a row's box list ends at its first polygon without a valid point."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    pt_off = np.ascontiguousarray(pt_off, dtype=np.int32)
    box_off = np.ascontiguousarray(box_off, dtype=np.int32)
    n, nb = len(box_off) - 1, len(pt_off) - 1
    if n < 0 or nb < 0 or box_off[0] != 0 or pt_off[0] != 0 or int(box_off[-1]) != nb or 2 * int(pt_off[-1]) != xy.size:
        raise ValueError("box_off / pt_off / xy sizes disagree")
    arg = np.empty((nb, 4), np.int32)
    high = np.zeros(n, np.uint8)
    box = np.empty((nb, 4), np.float64) if want_box else None
    check(lib().dyd_bbox_iou_fused(_ptr(xy), _ptr(pt_off), _ptr(box_off), n, int(min_boxes), float(thr),
                                   _ptr(box) if want_box else None, _ptr(arg), _ptr(high)), "dyd_bbox_iou_fused")
    return (arg, high, box) if want_box else (arg, high)


def hash128(data: np.ndarray, off: np.ndarray) -> np.ndarray:
    """K3 over host arrays: [n,2] u64."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    n = len(off) - 1
    if n < 0 or off[0] != 0 or int(off[-1]) != data.size:
        raise ValueError("off must start at 0 and end at len(bytes)")
    out = np.empty((n, 2), np.uint64)
    check(lib().dyd_hash128(_ptr(data) if data.size else None, _ptr(off), n, _ptr(out)), "dyd_hash128")
    return out


def dedup(h: np.ndarray, keep) -> np.ndarray:
    """K4 over host arrays: keep-mask (uint8) for keep in {"first", "last", False}."""
    if keep not in _KEEP:
        raise ValueError('keep must be either "first", "last" or False')
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2)
    out = np.empty(len(h), np.uint8)
    check(lib().dyd_dedup(_ptr(h), len(h), _KEEP[keep], _ptr(out)), "dyd_dedup")
    return out


def isin(h: np.ndarray, ref_h: np.ndarray) -> np.ndarray:
    """K5 over host arrays."""
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2)
    ref_h = np.ascontiguousarray(ref_h, dtype=np.uint64).reshape(-1, 2)
    out = np.empty(len(h), np.uint8)
    check(lib().dyd_isin(_ptr(h), len(h), _ptr(ref_h) if len(ref_h) else None, len(ref_h), _ptr(out)), "dyd_isin")
    return out


def dedup_partner(h: np.ndarray) -> np.ndarray:
    """per row the first row whose hash equals its own (int64; the row itself for a first occurrence)"""
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2)
    out = np.empty(len(h), np.int64)
    check(lib().dyd_dedup_partner(_ptr(h), len(h), _ptr(out)), "dyd_dedup_partner")
    return out


def isin_partner(h: np.ndarray, ref_h: np.ndarray) -> np.ndarray:
    """per main row the reference row whose hash it equals (int64, -1: none)"""
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2)
    ref_h = np.ascontiguousarray(ref_h, dtype=np.uint64).reshape(-1, 2)
    out = np.empty(len(h), np.int64)
    check(lib().dyd_isin_partner(_ptr(h), len(h), _ptr(ref_h) if len(ref_h) else None, len(ref_h), _ptr(out)), "dyd_isin_partner")
    return out


def cells_differ(text_a, off_a, idx_a, text_b, off_b, idx_b, n: int) -> np.ndarray:
    """uint8 per pair: do the cells a[idx_a[i]] and b[idx_b[i]] (flat utf-8 + int64 offsets) differ?  Pairs with a negative index
    count as equal.  Host code inside the library, multithreaded; needs no GPU."""
    text_a = np.ascontiguousarray(text_a, dtype=np.uint8); text_b = np.ascontiguousarray(text_b, dtype=np.uint8)
    off_a = np.ascontiguousarray(off_a, dtype=np.int64); off_b = np.ascontiguousarray(off_b, dtype=np.int64)
    idx_a = None if idx_a is None else np.ascontiguousarray(idx_a, dtype=np.int64)
    idx_b = None if idx_b is None else np.ascontiguousarray(idx_b, dtype=np.int64)
    out = np.zeros(int(n), np.uint8)
    if n:
        load_library().dyd_host_cells_differ(_ptr(text_a) if text_a.size else None, _ptr(off_a), None if idx_a is None else _ptr(idx_a),
                                             _ptr(text_b) if text_b.size else None, _ptr(off_b), None if idx_b is None else _ptr(idx_b),
                                             int(n), 0, _ptr(out))
    return out


def mt19937_permutation(seed: int, n: int) -> np.ndarray:
    """Host code inside the library: numpy legacy RandomState(seed).permutation(n)."""
    if not (0 <= int(seed) <= 2 ** 32 - 1):
        raise ValueError("Seed must be between 0 and 2**32 - 1")   # numpy's message (mtrand legacy seeding)
    out = np.empty(int(n), np.int64)
    check(load_library().dyd_mt19937_permutation(int(seed), int(n), _ptr(out)), "dyd_mt19937_permutation")
    return out


def mt19937_permutation_device(seed: int, n: int, want_inverse: bool = False):
    """K8: the same permutation computed in parallel on the GPU (n <= 2^30) -> perm [, inverse] as int64 host arrays"""
    if not (0 <= int(seed) <= 2 ** 32 - 1):
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    L = lib()
    n = int(n)
    nbytes = 8 * max(n, 1)
    d_perm, d_inv = C.c_void_p(), C.c_void_p()
    check(L.dyd_malloc(C.byref(d_perm), nbytes), "dyd_malloc")
    check(L.dyd_malloc(C.byref(d_inv), nbytes), "dyd_malloc")
    try:
        check(L.dyd_mt19937_permutation_dev(int(seed), n, d_perm, d_inv, None), "dyd_mt19937_permutation_dev")
        perm, inv = np.empty(n, np.int64), np.empty(n, np.int64)
        if n:
            check(L.dyd_d2h(_ptr(perm), d_perm, 8 * n), "dyd_d2h")
            check(L.dyd_d2h(_ptr(inv), d_inv, 8 * n), "dyd_d2h")
    finally:
        L.dyd_free(d_perm)
        L.dyd_free(d_inv)
    return (perm, inv) if want_inverse else perm


def split_ids_seeded(cat, seed: int, sizes, n_train, n_val):
    """K8 + K6 over host arrays: (split u8, pos i64); the categories' permutations are made on the device from `seed`"""
    if not (0 <= int(seed) <= 2 ** 32 - 1):
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    cat = np.ascontiguousarray(cat, dtype=np.int32)
    sizes = np.ascontiguousarray(sizes, dtype=np.int64)
    n_train = np.ascontiguousarray(n_train, dtype=np.int64)
    n_val = np.ascontiguousarray(n_val, dtype=np.int64)
    if not (len(sizes) == len(n_train) == len(n_val)):
        raise ValueError("sizes / n_train / n_val sizes disagree")
    split = np.empty(len(cat), np.uint8)
    pos = np.empty(len(cat), np.int64)
    check(lib().dyd_split_ids_seeded(_ptr(cat), len(cat), int(seed), _ptr(sizes), _ptr(n_train), _ptr(n_val), len(sizes),
                                     _ptr(split), _ptr(pos)), "dyd_split_ids_seeded")
    return split, pos


def split_ids(cat, perm_concat, cat_off, n_train, n_val):
    """K6 over host arrays: (split u8, pos i64)."""
    cat = np.ascontiguousarray(cat, dtype=np.int32)
    perm_concat = np.ascontiguousarray(perm_concat, dtype=np.int64)
    cat_off = np.ascontiguousarray(cat_off, dtype=np.int64)
    n_train = np.ascontiguousarray(n_train, dtype=np.int64)
    n_val = np.ascontiguousarray(n_val, dtype=np.int64)
    n_cat = len(n_train)
    if len(cat_off) != n_cat + 1 or len(n_val) != n_cat or int(cat_off[-1]) != len(perm_concat):
        raise ValueError("cat_off / n_train / n_val / perm sizes disagree")
    split = np.empty(len(cat), np.uint8)
    pos = np.empty(len(cat), np.int64)
    check(lib().dyd_split_ids(_ptr(cat), len(cat), _ptr(perm_concat), _ptr(cat_off), _ptr(n_train), _ptr(n_val),
                              n_cat, _ptr(split), _ptr(pos)), "dyd_split_ids")
    return split, pos


def yolo_lines(box4, row_off, sel, width, height, class_id):
    """K7 over host arrays -> (text_off int64 [n+1], flag u8 [n], text bytes).  flag 2 rows carry no text:
    the caller prints them (zero image size, values of 2^43 and more)."""
    box4 = np.ascontiguousarray(box4, dtype=np.float64).reshape(-1)
    row_off = np.ascontiguousarray(row_off, dtype=np.int32)
    n = len(row_off) - 1
    width = np.ascontiguousarray(width, dtype=np.float64)
    height = np.ascontiguousarray(height, dtype=np.float64)
    class_id = np.ascontiguousarray(class_id, dtype=np.int32)
    if n < 0 or len(width) != n or len(height) != n or len(class_id) != n:
        raise ValueError("row_off / width / height / class_id sizes disagree")
    if n and (int(row_off[-1]) * 4 != len(box4)):
        raise ValueError("row_off[-1] != number of boxes")
    sel_p = None
    if sel is not None:
        sel = np.ascontiguousarray(sel, dtype=np.uint8)
        if len(sel) * 4 != len(box4):
            raise ValueError("sel size != number of boxes")
        sel_p = _ptr(sel)
    off = np.zeros(n + 1, np.int64)
    flag = np.zeros(n, np.uint8)
    text, total = C.c_void_p(), C.c_int64()
    L = lib()
    check(L.dyd_yolo_lines(_ptr(box4), _ptr(row_off), sel_p, _ptr(width), _ptr(height), _ptr(class_id), n, _ptr(off),
                           _ptr(flag), C.byref(text), C.byref(total)), "dyd_yolo_lines")
    try:
        data = C.string_at(text.value, total.value) if total.value else b""
    finally:
        L.dyd_host_free(text)
    return off, flag, data
