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


MIN_THREADED = 65536          # below this many elements the wrappers leave a gather to numpy (see set_min_threaded)


def set_min_threaded(n: int = 65536) -> None:
    """tests: run the threaded builders from ``n`` elements on (default 65536) — here and inside the extension"""
    global MIN_THREADED
    MIN_THREADED = n
    if available():
        _dydpy.set_min_parallel(-1 if n == 65536 else n)


def _threads(n_threads: int) -> int:
    from . import native_json as _nj

    return n_threads or _nj.host_threads()


def _checked_index(idx, size) -> np.ndarray:
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    if len(idx) and (int(idx.min()) < 0 or int(idx.max()) >= size):
        raise IndexError("index out of range")
    return idx


def _addr(a) -> int:
    return 0 if a is None else a.ctypes.data


# The column builders below all compute   out[slot[i]] = f(idx[i])   (either index optional) on worker threads: the split step
# walks its records in row order — sequential reads, a source row's objects many times in a row — and scatters to the place the
# shuffle gave each record.  ``slot`` must be a permutation of range(n) (every output element written once); ``checked=True``
# says the index arrays were made by this package and need no range check.


def strings_from_views(ptr: np.ndarray, length: np.ndarray, idx=None, n_threads: int = 0, all_ascii: bool = False, slot=None,
                       checked: bool = False, na=None) -> np.ndarray:
    """object array of str from one (address, length) view per text: out[slot[i]] = text idx[i] — the split step's records,
    straight from the native handle's buffers (no flat copy of the text in either order).
    ``all_ascii``: the caller vouches that every text is ASCII (saves the classifying pass); ``na[i] != 0`` leaves out[slot[i]] None."""
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.int64)
    n = len(ptr) if idx is None else len(idx)
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int64) if checked else _checked_index(idx, len(ptr))
    if slot is not None:
        slot = np.ascontiguousarray(slot, dtype=np.int64) if checked else _checked_index(slot, n)
        if len(slot) != n:
            raise ValueError("slot and idx differ in length")
    if na is not None:
        na = np.ascontiguousarray(na, dtype=np.uint8)
        if len(na) != n:
            raise ValueError("na and the walk differ in length")
    out = np.empty(n, object)
    if n:
        _dydpy.map_strs(ptr.ctypes.data, length.ctypes.data, _addr(idx), _addr(slot), n, out.ctypes.data, _threads(n_threads),
                        1 if all_ascii else 0, _addr(na))
    return out


def alloc_strings(ptr: np.ndarray, length: np.ndarray, all_ascii: bool = False, n_threads: int = 0):
    """First half of strings_from_views for callers that learn the places later: -> (seq, ascii) where seq[i] is an ASCII str of
    length[i] characters whose text is still UNWRITTEN (or the finished str when text i is not ASCII; ascii[i] says which; ascii
    is None with ``all_ascii``).  The strings must not be looked at before fill_strings has written them."""
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.int64)
    n = len(ptr)
    seq = np.empty(n, object)
    ascii_ = None if all_ascii else np.zeros(n, np.uint8)
    if n:
        _dydpy.alloc_strs(ptr.ctypes.data, length.ctypes.data, n, seq.ctypes.data, _threads(n_threads), 1 if all_ascii else 0, _addr(ascii_))
    return seq, ascii_


def fill_strings(ptr: np.ndarray, length: np.ndarray, seq: np.ndarray, ascii_=None, slot=None, out=None, n_threads: int = 0) -> np.ndarray:
    """Second half: the texts are written into alloc_strings' strings by worker threads (no GIL) and — with ``out`` — every string
    MOVES to out[slot[i]] (seq is left holding None).  Without ``out`` the strings are filled in place and seq is returned."""
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.int64)
    n = len(seq)
    if len(ptr) != n or len(length) != n or (slot is not None and len(slot) != n):
        raise ValueError("fill_strings: lengths differ")
    if slot is not None:
        slot = np.ascontiguousarray(slot, dtype=np.int64)
        if out is None or (n and (int(slot.min()) < 0 or int(slot.max()) >= len(out))):
            raise IndexError("fill_strings: slot out of range")
    if n:
        _dydpy.fill_strs(ptr.ctypes.data, length.ctypes.data, n, seq.ctypes.data, _addr(slot), _addr(out), _threads(n_threads), _addr(ascii_))
    return seq if out is None else out


def take(values: np.ndarray, idx=None, n_threads: int = 0, checked: bool = False, slot=None) -> np.ndarray:
    """out[slot[i]] = values[idx[i]] on worker threads: 1-D object arrays (references counted with one atomic add per run of equal
    objects — numpy walks the scattered object headers on one thread) and 1-D arrays of 1 / 2 / 4 / 8-byte items; anything else,
    and short index arrays, through numpy."""
    n = len(values) if idx is None else len(idx)
    if (not available() or not isinstance(values, np.ndarray) or values.ndim != 1 or n < MIN_THREADED
            or not (values.dtype == object or (values.dtype.kind in "iufb" and values.dtype.itemsize in (1, 2, 4, 8)))):
        got = values if idx is None else values[idx]
        if slot is None:
            return got
        out = np.empty(n, got.dtype)
        out[slot] = got
        return out
    values = np.ascontiguousarray(values)
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int64) if checked else _checked_index(idx, len(values))
    if slot is not None:
        slot = np.ascontiguousarray(slot, dtype=np.int64) if checked else _checked_index(slot, n)
        if len(slot) != n:
            raise ValueError("slot and idx differ in length")
    out = np.empty(n, values.dtype)
    if values.dtype == object:
        _dydpy.map_objects(values.ctypes.data, _addr(idx), _addr(slot), n, out.ctypes.data, _threads(n_threads))
    else:
        _dydpy.map_fixed(values.ctypes.data, values.dtype.itemsize, _addr(idx), _addr(slot), n, out.ctypes.data, _threads(n_threads))
    return out


def take_small(table: np.ndarray, codes: np.ndarray, idx=None, n_threads: int = 0, slot=None, checked: bool = False) -> np.ndarray:
    """out[slot[i]] = table[codes[idx[i]]] for a SMALL object table (the labels of the rules) and int32 codes: every worker adds
    its uses of an entry to the reference count once instead of once per element"""
    n = len(codes) if idx is None else len(idx)
    if not available() or n < MIN_THREADED or len(table) > 65536:
        got = table[codes if idx is None else codes[idx]]
        if slot is None:
            return got
        out = np.empty(n, object)
        out[slot] = got
        return out
    table = np.ascontiguousarray(table, dtype=object)
    codes = np.ascontiguousarray(codes, dtype=np.int32)
    if len(codes) and (int(codes.min()) < 0 or int(codes.max()) >= len(table)):
        raise IndexError("take_small: code out of range")
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int64) if checked else _checked_index(idx, len(codes))
    if slot is not None:
        slot = np.ascontiguousarray(slot, dtype=np.int64) if checked else _checked_index(slot, n)
    out = np.empty(n, object)
    _dydpy.map_small(table.ctypes.data, len(table), codes.ctypes.data, _addr(idx), _addr(slot), n, out.ctypes.data, _threads(n_threads))
    return out


def category_slots(cat: np.ndarray, pos: np.ndarray, cat_off: np.ndarray, n_threads: int = 0) -> np.ndarray:
    """slot[e] = cat_off[cat[e]] + pos[e]: where record e stands once the categories are laid out one after the other, each in
    its shuffled order (K6's positions are a permutation inside each category, so slot is a permutation of range(n))"""
    n = len(cat)
    if not available() or n < MIN_THREADED:
        slot = cat_off[cat].astype(np.int64) if n else np.zeros(0, np.int64)
        slot += pos
        return slot
    out = np.empty(n, np.int64)
    cat = np.ascontiguousarray(cat, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    cat_off = np.ascontiguousarray(cat_off, dtype=np.int64)
    if int(cat.min()) < 0 or int(cat.max()) >= len(cat_off) - 1:
        raise IndexError("category_slots: category out of range")
    _dydpy.category_slots(cat.ctypes.data, pos.ctypes.data, cat_off.ctypes.data, n, out.ctypes.data, _threads(n_threads))
    return out


def gather_text(ptr: np.ndarray, length: np.ndarray, idx=None, n_threads: int = 0, slot=None, checked: bool = False):
    """(contiguous utf-8 buffer u8, offsets i64 [n+1]) with text idx[i] at place slot[i], the texts given as (address, length) views"""
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.int64)
    n = len(ptr) if idx is None else len(idx)
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int64) if checked else _checked_index(idx, len(ptr))
    if slot is not None:
        slot = np.ascontiguousarray(slot, dtype=np.int64) if checked else _checked_index(slot, n)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(take(length, idx, checked=True, slot=slot), out=off[1:])
    data = np.empty(max(int(off[-1]), 1), np.uint8)
    if n:
        _dydpy.map_text(ptr.ctypes.data, length.ctypes.data, _addr(idx), _addr(slot), n, off.ctypes.data, data.ctypes.data,
                        _threads(n_threads))
    return data[:int(off[-1])], off


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
