"""ctypes wrappers over liboracle.so (oracle/dyd_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, see oracle/Makefile)."""
    src = os.path.join(_HERE, "dyd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def bbox_minmax(xy: np.ndarray, pt_off: np.ndarray):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    pt_off = np.ascontiguousarray(pt_off, dtype=np.int32)
    nb = len(pt_off) - 1
    box = np.empty((nb, 4), np.float64)
    arg = np.empty((nb, 4), np.int32)
    _load().orc_bbox_minmax(_p(xy, C.c_double), _p(pt_off, C.c_int32), C.c_int64(nb),
                            _p(box, C.c_double), _p(arg, C.c_int32))
    return box, arg


def iou_any_ge(box4: np.ndarray, row_off: np.ndarray, min_boxes: int, thr: float, want_max=False):
    box4 = np.ascontiguousarray(box4, dtype=np.float64).reshape(-1)
    row_off = np.ascontiguousarray(row_off, dtype=np.int32)
    n = len(row_off) - 1
    high = np.empty(n, np.uint8)
    mx = np.empty(n, np.float64) if want_max else None
    _load().orc_iou_any_ge(_p(box4, C.c_double), _p(row_off, C.c_int32), C.c_int64(n),
                           C.c_int32(min_boxes), C.c_double(thr), _p(high, C.c_uint8),
                           _p(mx, C.c_double) if want_max else None)
    return (high, mx) if want_max else high


def bbox_iou_chain(xy: np.ndarray, pt_off: np.ndarray, box_off: np.ndarray, min_boxes: int, thr: float):
    """The reference's replace step followed by its IoU step on the same rows (processor.py:262-281 then :341-376), at the
    array interface: K1 per polygon, then per row the all-pairs test over the boxes BEFORE the row's first polygon without a
    valid point — such a polygon is emitted with null coordinates (:254-255), min(None, None) raises inside extract_boxes'
    blanket try (:359, :364-365) and the boxes collected so far are what meet_conditions sees.  -> (box4, arg4, high)"""
    box, arg = bbox_minmax(xy, pt_off)
    box_off = np.ascontiguousarray(box_off, dtype=np.int64)
    n = len(box_off) - 1
    empty = np.flatnonzero(arg[:, 0] < 0)
    cnt = np.diff(box_off)
    if len(empty):
        row = np.searchsorted(box_off, empty, side="right") - 1
        first = np.full(n, np.iinfo(np.int64).max)
        np.minimum.at(first, row, empty - box_off[row])
        cnt = np.minimum(cnt, first)
    off2 = np.zeros(n + 1, np.int64)
    np.cumsum(cnt, out=off2[1:])
    keep = np.repeat(box_off[:-1] - off2[:-1], cnt) + np.arange(int(off2[-1]))
    high = iou_any_ge(box[keep], off2.astype(np.int32), min_boxes, thr)
    return box, arg, high


def hash128(data: np.ndarray, off: np.ndarray):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if data.size == 0:
        data = np.zeros(1, np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    n = len(off) - 1
    out = np.empty((n, 2), np.uint64)
    _load().orc_hash128(_p(data, C.c_uint8), _p(off, C.c_int64), C.c_int64(n), _p(out, C.c_uint64))
    return out


def dedup(h: np.ndarray, keep_mode: int):
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2)
    out = np.empty(len(h), np.uint8)
    _load().orc_dedup(_p(h, C.c_uint64), C.c_int64(len(h)), C.c_int(keep_mode), _p(out, C.c_uint8))
    return out


def isin(h: np.ndarray, ref_h: np.ndarray):
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2)
    ref_h = np.ascontiguousarray(ref_h, dtype=np.uint64).reshape(-1, 2)
    out = np.empty(len(h), np.uint8)
    _load().orc_isin(_p(h, C.c_uint64), C.c_int64(len(h)), _p(ref_h, C.c_uint64),
                     C.c_int64(len(ref_h)), _p(out, C.c_uint8))
    return out


def mt19937_permutation(seed: int, n: int):
    out = np.empty(n, np.int64)
    _load().orc_mt19937_permutation(C.c_uint32(seed), C.c_int64(n), _p(out, C.c_int64))
    return out


def split_ids(cat, perm_concat, cat_off, n_train, n_val):
    cat = np.ascontiguousarray(cat, dtype=np.int32)
    perm_concat = np.ascontiguousarray(perm_concat, dtype=np.int64)
    cat_off = np.ascontiguousarray(cat_off, dtype=np.int64)
    n_train = np.ascontiguousarray(n_train, dtype=np.int64)
    n_val = np.ascontiguousarray(n_val, dtype=np.int64)
    n = len(cat)
    split = np.empty(n, np.uint8)
    pos = np.empty(n, np.int64)
    _load().orc_split_ids(_p(cat, C.c_int32), C.c_int64(n), _p(perm_concat, C.c_int64),
                          _p(cat_off, C.c_int64), _p(n_train, C.c_int64), _p(n_val, C.c_int64),
                          C.c_int32(len(n_train)), _p(split, C.c_uint8), _p(pos, C.c_int64))
    return split, pos


def yolo_lines(box4, row_off, sel, width, height, class_id):
    """-> (text_off int64 [n+1], flag u8 [n], text bytes)"""
    lib = _load()
    box4 = np.ascontiguousarray(box4, dtype=np.float64).reshape(-1)
    row_off = np.ascontiguousarray(row_off, dtype=np.int32)
    n = len(row_off) - 1
    width = np.ascontiguousarray(width, dtype=np.float64)
    height = np.ascontiguousarray(height, dtype=np.float64)
    class_id = np.ascontiguousarray(class_id, dtype=np.int32)
    sel_p = None
    if sel is not None:
        sel = np.ascontiguousarray(sel, dtype=np.uint8)
        sel_p = _p(sel, C.c_uint8)
    off = np.zeros(n + 1, np.int64)
    flag = np.zeros(n, np.uint8)
    fn = lib.orc_yolo_lines
    fn.restype = C.c_int64
    args = (_p(box4, C.c_double), _p(row_off, C.c_int32), sel_p, _p(width, C.c_double), _p(height, C.c_double),
            _p(class_id, C.c_int32), C.c_int64(n), _p(off, C.c_int64), _p(flag, C.c_uint8))
    total = fn(*args, None)
    text = np.zeros(max(int(total), 1), np.uint8)
    fn(*args, _p(text, C.c_uint8))
    return off, flag, text[:total].tobytes()
