"""pandas object column <-> flat UTF-8 buffers without per-cell Python work.

``views`` hands the native JSON scanner the address and length of every str cell's UTF-8 form (no copy: the ASCII
text of a str object is its own buffer); ``strings`` turns emitted UTF-8 text + offsets back into str objects, the
bytes copied by worker threads.  Both live in the small CPython extension ``_dydpy`` (csrc/pyhelpers.c); when that
module is not built for the running interpreter the callers use the portable paths (native_json.cells_to_buffers,
pyarrow's to_pylist), which give the same results more slowly.  Host glue only — nothing here computes."""
from __future__ import annotations

import numpy as np

try:
    from . import _dydpy
except ImportError:                                    # not built for this interpreter
    _dydpy = None


def available() -> bool:
    return _dydpy is not None


class CellViews:
    """UTF-8 views of an object array's str cells.  Keeps the array (and so the str objects) alive."""

    def __init__(self, cells):
        arr = cells if isinstance(cells, np.ndarray) and cells.dtype == object else None
        if arr is None:
            arr = np.empty(len(cells), object)
            arr[:] = cells if not isinstance(cells, np.ndarray) else cells.tolist()
        arr = np.ascontiguousarray(arr)
        n = len(arr)
        self.cells = arr
        self.ptr = np.empty(n, np.uint64)
        self.len = np.empty(n, np.int64)
        self.missing = np.empty(n, np.uint8)
        if n:
            from .native_json import host_threads
            _dydpy.str_views(arr.ctypes.data, n, self.ptr.ctypes.data, self.len.ctypes.data, self.missing.ctypes.data, host_threads())

    def __len__(self):
        return len(self.cells)


def all_str(arr: np.ndarray) -> bool:
    """every element of the object array is an exact str (then the column holds no missing value): a parallel walk of the object
    headers instead of ``Series.isna``'s per-object one.  False when the extension is missing or the array is not an object array."""
    if not available() or not isinstance(arr, np.ndarray) or arr.dtype != object or arr.ndim != 1:
        return False
    arr = np.ascontiguousarray(arr)
    if not len(arr):
        return True
    from .native_json import host_threads
    return bool(_dydpy.all_exact_str(arr.ctypes.data, len(arr), host_threads()))


def strings(text: np.ndarray, off: np.ndarray, na=None, n_threads: int = 0) -> np.ndarray:
    """object array of str (None where na != 0) from flat UTF-8 ``text`` and int64 ``off`` [n+1]"""
    from . import native_json as _nj

    n = len(off) - 1
    out = np.empty(n, object)
    if n == 0:
        return out
    text = np.ascontiguousarray(text, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    na_arr = None if na is None else np.ascontiguousarray(na, dtype=np.uint8)
    _dydpy.strs_from_utf8(text.ctypes.data if text.size else 0, off.ctypes.data, n, 0 if na_arr is None else na_arr.ctypes.data,
                          out.ctypes.data, n_threads or _nj.host_threads())
    return out


def flat_utf8(col_values: np.ndarray, na: np.ndarray, na_as_text: bool = False):
    """(flat utf-8 bytes u8, offsets i64 [n+1]) of an object array whose present cells are ALL str — or None when some present
    cell is something else (the caller's tagged spelling handles those).  Missing cells come out empty, or — ``na_as_text``,
    for ``astype(str)`` — as str(cell): "nan" for NaN, "None" for None.  The str objects' own UTF-8 buffers are copied once, by
    worker threads."""
    import ctypes as C

    from . import native_json as _nj

    v = CellViews(col_values)
    if len(v) and (v.missing.astype(bool) != na).any():
        return None
    lens, ptr, keep = v.len, v.ptr, {}
    if na_as_text and na.any():
        lens = lens.copy(); ptr = ptr.copy()
        for i in np.flatnonzero(na).tolist():
            t = str(v.cells[i])
            if t not in keep:
                keep[t] = t.encode("utf-8")
            b = keep[t]
            lens[i] = len(b)
            ptr[i] = C.cast(C.c_char_p(b), C.c_void_p).value
    off = np.zeros(len(v) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    data = np.empty(max(int(off[-1]), 1), np.uint8)
    if len(v):
        _dydpy.gather_utf8(ptr.ctypes.data, np.ascontiguousarray(lens).ctypes.data, off.ctypes.data, len(v), data.ctypes.data,
                           _nj.host_threads())
    del keep
    return data[:int(off[-1])], off
