"""Python face of the native flatten / emit (csrc/host_json.cpp).

``scan_polygons`` / ``scan_boxes`` turn a column of annotation cells into the SoA buffers of K1 / K2
without CPython's ``json``; ``PolygonScan.emit`` writes the rewritten JSON text.  Cells the native
scanner classifies as irregular (status 2) are NOT processed here: the callers in core/processor.py
run them through flatten.py, which follows the reference accessor by accessor.

This is host code inside libdyd_gfx950.so: it needs the library but no GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native

OK, UNDECODABLE, IRREGULAR, MISSING = 0, 1, 2, 3


def enabled() -> bool:
    return os.environ.get("DYD_NATIVE_JSON", "1") != "0"


_THREADS = None


def host_threads() -> int:
    """worker threads for host-side passes: DYD_HOST_THREADS, else the process's CPU share (cgroup quota when there is one —
    a GPU box hands a 16-CPU slice of a 256-thread host to one GPU), at most 64"""
    global _THREADS
    if _THREADS is None:
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, round(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
        env = os.environ.get("DYD_HOST_THREADS")
        if env and env.isdigit() and int(env) > 0:
            n = int(env)
        _THREADS = max(1, min(n, 64))
    return _THREADS


def cells_to_buffers(cells):
    """list / Series of cells -> (utf-8 bytes u8, offsets i64, missing u8, keep-alive object).

    Non-str cells count as missing (reference processor.py:264, :344).  Cells are joined with one
    blank (legal trailing JSON whitespace) and encoded in ONE pass; byte offsets follow from the
    str lengths, re-measured only for cells that are not pure ASCII.  Raises UnicodeEncodeError for
    lone surrogates (the caller then takes the Python path, which reproduces to_csv's failure)."""
    vals = [c if type(c) is str else "" for c in cells]
    n = len(vals)
    missing = np.fromiter((type(c) is not str for c in cells), dtype=np.uint8, count=n)
    blob = " ".join(vals)
    data = blob.encode("utf-8")
    lens = np.fromiter(map(len, vals), dtype=np.int64, count=n)
    if len(data) != len(blob):                     # some cell holds non-ASCII text: measure those in bytes
        for i, v in enumerate(vals):
            if not v.isascii():
                lens[i] = len(v.encode("utf-8"))
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens + 1, out=off[1:])               # +1: the separator blank belongs to the cell before it
    if n:
        off[-1] -= 1                               # no blank after the last cell
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return buf, off, missing, data


def _view(ptr, dtype, count):
    if count == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))


class _Scan:
    def __init__(self, handle, n_cells, keep):
        self._h = handle
        self._keep = keep            # arrow array / numpy buffers the native side points into
        L = _native.load_library()
        self.n_cells = n_cells
        self.n_boxes = int(L.dyd_scan_n_boxes(handle))
        self.status = _view(L.dyd_scan_status(handle), np.uint8, n_cells).copy()
        self.cell_box_off = _view(L.dyd_scan_cell_box_off(handle), np.int32, n_cells + 1).copy()

    def close(self):
        if self._h:
            _native.load_library().dyd_scan_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class PolygonScan(_Scan):
    """Result of scan_polygons: SoA points of the regular cells + what emit needs."""

    def __init__(self, handle, n_cells, keep, data, off):
        super().__init__(handle, n_cells, keep)
        L = _native.load_library()
        n_pts = int(L.dyd_scan_n_points(handle))
        self.xy = _view(L.dyd_scan_xy(handle), np.float64, 2 * n_pts).reshape(-1, 2)   # views into the handle
        self.pt_off = _view(L.dyd_scan_pt_off(handle), np.int32, self.n_boxes + 1)
        self._data, self._off = data, off
        self.w_kind = _view(L.dyd_scan_wh_kind(handle, 0), np.uint8, n_cells).copy()
        self.h_kind = _view(L.dyd_scan_wh_kind(handle, 1), np.uint8, n_cells).copy()
        self.w_val = _view(L.dyd_scan_wh_value(handle, 0), np.float64, n_cells).copy()
        self.h_val = _view(L.dyd_scan_wh_value(handle, 1), np.float64, n_cells).copy()
        self.iou_host = _view(L.dyd_scan_iou_host(handle), np.uint8, n_cells).copy()   # cells whose IoU flag the host decides
        self.fast_cells = int(L.dyd_scan_fast_cells(handle))                           # cells the single-parse lane took

    def emit_buffers(self, arg4: np.ndarray, n_threads: int = 0) -> tuple:
        """Rewritten JSON text of every cell as flat utf-8 + offsets (views into the handle, valid until
        close()); cells that are not regular have empty text."""
        L = _native.load_library()
        arg4 = np.ascontiguousarray(arg4, dtype=np.int32).reshape(-1)
        if arg4.size != 4 * self.n_boxes:
            raise ValueError("arg4 does not match the scan")
        out_text, out_off = C.c_void_p(), C.c_void_p()
        _native.check(L.dyd_json_emit_polygons(self._h, self._data.ctypes.data if self._data is not None else None,
                                               self._off.ctypes.data if self._off is not None else None,
                                               arg4.ctypes.data if arg4.size else None, n_threads,
                                               C.byref(out_text), C.byref(out_off)), "dyd_json_emit_polygons")
        off = _view(out_off.value, np.int64, self.n_cells + 1)
        return _view(out_text.value, np.uint8, int(off[-1])), off

    def emit_array(self, arg4: np.ndarray, n_threads: int = 0) -> np.ndarray:
        """object array: rewritten JSON text (str) for regular cells, None for every other cell"""
        text, off = self.emit_buffers(arg4, n_threads)
        return strings_from_buffers(text, off, (self.status != OK).astype(np.uint8))

    def emit(self, arg4: np.ndarray, n_threads: int = 0) -> list:
        """Rewritten JSON text per cell: str for regular cells, None for undecodable / missing cells and
        for irregular ones (the caller fills those in)."""
        return self.emit_array(arg4, n_threads).tolist()

    def wh_column(self, which: int):
        """doc.get("width") / doc.get("height") of every cell as ONE column value for ``frame[col] = ...``: a numpy array
        when the cells hold only ints / floats / nothing (int64 when all are ints, float64 with NaN otherwise — the dtypes
        pandas infers from the reference's list, processor.py:295-296), else the per-cell list of width_height()."""
        kind = self.w_kind if which == 0 else self.h_kind
        val = self.w_val if which == 0 else self.h_val
        if self.n_cells == 0 or (kind == 3).any() or not kind.any():
            return self.width_height(which)
        if (kind == 1).all():
            return val.astype(np.int64)
        out = val.copy()
        out[kind == 0] = np.nan
        return out

    def width_height(self, which: int) -> list:
        """Python values of doc.get("width") / doc.get("height") for regular cells (None elsewhere);
        kind 3 (string / container / huge int) is returned as the marker ``Ellipsis`` for the caller."""
        kind = self.w_kind if which == 0 else self.h_kind
        val = self.w_val if which == 0 else self.h_val
        out = [None] * self.n_cells
        for i in np.flatnonzero(kind).tolist():
            k = kind[i]
            out[i] = int(val[i]) if k == 1 else (float(val[i]) if k == 2 else Ellipsis)
        return out


class ReplaceIou:
    """Result of the native replace -> IoU pipeline (dyd_json_replace_iou): per-cell status / HIGH flag / width / height and
    the emitted bbox text, per part or gathered.  Cells with status IRREGULAR, and cells with ``iou_host`` set, are the caller's."""

    def __init__(self, handle, n_cells, keep):
        L = _native.load_library()
        self._h, self._keep, self.n_cells = handle, keep, n_cells
        self.status = _view(L.dyd_scan_status(handle), np.uint8, n_cells).copy()
        self.high = _view(L.dyd_scan_high(handle), np.uint8, n_cells).astype(bool)
        self.iou_host = _view(L.dyd_scan_iou_host(handle), np.uint8, n_cells).copy()
        self.w_kind = _view(L.dyd_scan_wh_kind(handle, 0), np.uint8, n_cells).copy()
        self.h_kind = _view(L.dyd_scan_wh_kind(handle, 1), np.uint8, n_cells).copy()
        self.w_val = _view(L.dyd_scan_wh_value(handle, 0), np.float64, n_cells).copy()
        self.h_val = _view(L.dyd_scan_wh_value(handle, 1), np.float64, n_cells).copy()
        counts, secs = np.zeros(3, np.int64), np.zeros(3, np.float64)
        L.dyd_scan_totals(handle, counts.ctypes.data, secs.ctypes.data)
        self.n_boxes, self.n_points, self.fast_cells = (int(v) for v in counts)
        self.n_parts = int(L.dyd_scan_parts(handle))          # = fused launches: one per worker thread's share of the cells
        self.seconds = {"scan": float(secs[0]), "device": float(secs[1]), "emit": float(secs[2])}

    wh_column = PolygonScan.wh_column
    width_height = PolygonScan.width_height

    def texts_array(self) -> np.ndarray:
        """object array: bbox text (str) for regular cells, None elsewhere — str objects made straight from the parts' buffers
        (no gathered copy of the text), all parts in ONE pass of the str builder"""
        from . import pycells

        L = _native.load_library()
        na = (self.status != OK).astype(np.uint8)
        if not pycells.available():
            out = np.empty(self.n_cells, object)
        ptr = np.zeros(self.n_cells, np.uint64)
        length = np.zeros(self.n_cells, np.int64)
        for k in range(int(L.dyd_scan_parts(self._h))):
            lo, hi, text, off = C.c_int64(), C.c_int64(), C.c_void_p(), C.c_void_p()
            _native.check(L.dyd_scan_part(self._h, k, C.byref(lo), C.byref(hi), C.byref(text), C.byref(off)), "dyd_scan_part")
            n = hi.value - lo.value
            if n == 0:
                continue
            offs = _view(off.value, np.int64, n + 1)
            if pycells.available():
                np.add(offs[:-1], text.value or 0, out=ptr[lo.value:hi.value], casting="unsafe")
                np.subtract(offs[1:], offs[:-1], out=length[lo.value:hi.value])
            else:
                out[lo.value:hi.value] = strings_from_buffers(_view(text.value, np.uint8, int(offs[-1])), offs, na[lo.value:hi.value])
        if pycells.available():
            return pycells.strings_from_views(ptr, length, na=na)
        return out

    def texts_arrow(self):
        """the bbox column as a pandas ArrowStringArray (dtype "string", pyarrow storage; missing where a cell is not regular)
        built straight ON the parts' buffers: no str objects, no copy.  The arrays keep this object — and so the native
        handle — alive; do not call close() while they are in use."""
        import pandas as pd
        import pyarrow as pa

        L = _native.load_library()
        chunks = []
        valid_all = self.status == OK
        for k in range(int(L.dyd_scan_parts(self._h))):
            lo, hi, text, off = C.c_int64(), C.c_int64(), C.c_void_p(), C.c_void_p()
            _native.check(L.dyd_scan_part(self._h, k, C.byref(lo), C.byref(hi), C.byref(text), C.byref(off)), "dyd_scan_part")
            n = hi.value - lo.value
            if n == 0:
                continue
            total = int(_view(off.value, np.int64, n + 1)[-1])
            valid = valid_all[lo.value:hi.value]
            bufs = [None if valid.all() else pa.py_buffer(np.packbits(valid, bitorder="little")),
                    pa.foreign_buffer(off.value, 8 * (n + 1), base=self),
                    pa.foreign_buffer(text.value, total, base=self) if total else pa.py_buffer(b"")]
            chunks.append(pa.LargeStringArray.from_buffers(n, bufs[1], bufs[2], bufs[0], -1 if bufs[0] is not None else 0))
        return pd.arrays.ArrowStringArray(pa.chunked_array(chunks, type=pa.large_string()))

    def text_buffers(self) -> tuple:
        """the emitted text of all cells as ONE flat utf-8 buffer + offsets (views into the handle)"""
        L = _native.load_library()
        text, off = C.c_void_p(), C.c_void_p()
        _native.check(L.dyd_scan_text(self._h, C.byref(text), C.byref(off)), "dyd_scan_text")
        offs = _view(off.value, np.int64, self.n_cells + 1)
        return _view(text.value, np.uint8, int(offs[-1])), offs

    def close(self):
        if self._h:
            _native.load_library().dyd_scan_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def replace_iou(cells, min_boxes: int, thr: float, n_threads: int = 0) -> ReplaceIou:
    """The native pipeline over a column of annotation cells (list / object ndarray / Series of str)."""
    from . import pycells

    L = _native.lib()                      # the pipeline launches kernels: needs the device
    h = C.c_void_p()
    if pycells.available():
        v = pycells.CellViews(cells.to_numpy() if hasattr(cells, "to_numpy") else cells)
        _native.check(L.dyd_json_replace_iou(None, None, v.ptr.ctypes.data, v.len.ctypes.data, v.missing.ctypes.data, len(v),
                                             int(min_boxes), float(thr), n_threads, C.byref(h)), "dyd_json_replace_iou")
        return ReplaceIou(h, len(v), v)
    data, off, missing, keep = cells_to_buffers(cells)
    _native.check(L.dyd_json_replace_iou(data.ctypes.data, off.ctypes.data, None, None, missing.ctypes.data, len(off) - 1,
                                         int(min_boxes), float(thr), n_threads, C.byref(h)), "dyd_json_replace_iou")
    return ReplaceIou(h, len(off) - 1, (keep, data, off, missing))


def replace_iou_buffers(data, off, missing, min_boxes: int, thr: float, n_threads: int = 0, keep=None) -> ReplaceIou:
    """the same over cells that already are flat utf-8 (fastcsv.Utf8Column)"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    missing = np.ascontiguousarray(missing, dtype=np.uint8)
    L = _native.lib()
    h = C.c_void_p()
    _native.check(L.dyd_json_replace_iou(data.ctypes.data, off.ctypes.data, None, None, missing.ctypes.data, len(off) - 1,
                                         int(min_boxes), float(thr), n_threads, C.byref(h)), "dyd_json_replace_iou")
    return ReplaceIou(h, len(off) - 1, (keep, data, off, missing))


class BoxScan(_Scan):
    def __init__(self, handle, n_cells, keep):
        super().__init__(handle, n_cells, keep)
        L = _native.load_library()
        nb = int(self.cell_box_off[-1]) if n_cells else 0
        self.box4 = _view(L.dyd_scan_xy(handle), np.float64, 4 * nb).reshape(-1, 4)
        self.row_off = self.cell_box_off


def scan_polygons_buffers(data, off, missing, n_threads: int = 0, keep=None) -> PolygonScan:
    """scan cells that already are flat utf-8 (fastcsv.Utf8Column): no Python str objects involved"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    missing = np.ascontiguousarray(missing, dtype=np.uint8)
    L = _native.load_library()
    h = C.c_void_p()
    _native.check(L.dyd_json_scan_polygons(data.ctypes.data, off.ctypes.data, missing.ctypes.data, len(off) - 1,
                                           n_threads, C.byref(h)), "dyd_json_scan_polygons")
    return PolygonScan(h, len(off) - 1, keep, data, off)


def scan_boxes_buffers(data, off, missing, n_threads: int = 0, keep=None) -> BoxScan:
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    missing = np.ascontiguousarray(missing, dtype=np.uint8)
    L = _native.load_library()
    h = C.c_void_p()
    _native.check(L.dyd_json_scan_boxes(data.ctypes.data, off.ctypes.data, missing.ctypes.data, len(off) - 1,
                                        n_threads, C.byref(h)), "dyd_json_scan_boxes")
    return BoxScan(h, len(off) - 1, (keep, data, off, missing))


def strings_from_buffers(text, off, na=None, n_threads: int = 0) -> np.ndarray:
    """object array of str (None where na) from flat utf-8 + offsets: the CPython helper when it is built, pyarrow otherwise"""
    from . import pycells

    n = len(off) - 1
    if pycells.available():
        return pycells.strings(text, off, na, n_threads)
    import pyarrow as pa

    arr = pa.LargeStringArray.from_buffers(n, pa.py_buffer(np.ascontiguousarray(off, dtype=np.int64)),
                                           pa.py_buffer(text if len(text) else b""))
    out = np.empty(n, object)
    out[:] = arr.to_pylist()
    if na is not None:
        out[np.asarray(na) != 0] = None
    return out


def scan_polygons(cells, n_threads: int = 0) -> PolygonScan:
    """cells: list / object ndarray / Series of annotation cells.  With the CPython helper the scanner reads the str objects'
    own UTF-8 buffers (no copy); otherwise the cells are joined and encoded first."""
    from . import pycells

    L = _native.load_library()
    h = C.c_void_p()
    if pycells.available():
        v = pycells.CellViews(cells.to_numpy() if hasattr(cells, "to_numpy") else cells)
        _native.check(L.dyd_json_scan_polygons_v(v.ptr.ctypes.data, v.len.ctypes.data, v.missing.ctypes.data, len(v),
                                                 n_threads, C.byref(h)), "dyd_json_scan_polygons_v")
        return PolygonScan(h, len(v), v, None, None)
    data, off, missing, keep = cells_to_buffers(cells)
    _native.check(L.dyd_json_scan_polygons(data.ctypes.data, off.ctypes.data, missing.ctypes.data, len(off) - 1,
                                           n_threads, C.byref(h)), "dyd_json_scan_polygons")
    return PolygonScan(h, len(off) - 1, keep, data, off)


def scan_boxes(cells, n_threads: int = 0) -> BoxScan:
    data, off, missing, keep = cells_to_buffers(cells)
    L = _native.load_library()
    h = C.c_void_p()
    _native.check(L.dyd_json_scan_boxes(data.ctypes.data, off.ctypes.data, missing.ctypes.data, len(off) - 1,
                                        n_threads, C.byref(h)), "dyd_json_scan_boxes")
    return BoxScan(h, len(off) - 1, keep)


# ------------------------------------------------------------------------------------------ split step
SP_OK, SP_EMPTY, SP_UNDECODABLE, SP_NOT_A_LIST, SP_NO_OBJECTS, SP_IRREGULAR = 0, 1, 2, 3, 4, 5
EV_NO_NAME, EV_UNDEFINED, EV_NOTHING_CLASSIFIED = 1, 2, 3


def _strings(data_ptr, off_ptr, count) -> np.ndarray:
    """object array of str from a flat utf-8 buffer + offsets (one C pass through pyarrow)"""
    import pyarrow as pa

    if count == 0:
        return np.empty(0, object)
    off = _view(off_ptr, np.int64, count + 1)
    data = _view(data_ptr, np.uint8, max(int(off[-1]), 1))
    arr = pa.LargeStringArray.from_buffers(count, pa.py_buffer(off), pa.py_buffer(data))
    return arr.to_numpy(zero_copy_only=False).astype(object)


class SplitExpansion:
    """Result of split_expand.  The fixed-width arrays are copied out; the record texts stay in the native handle's per-thread
    buffers (``rec_ptr`` / ``rec_len``: one view per record) until ``close()`` — ``record_strings`` turns them into str objects, in
    any order, without a flat copy of the text in between."""

    def __init__(self, handle, n_cells, string_threads: int = 0, late_text: bool = False):
        L = _native.load_library()
        self._h = handle
        self.n_cells = n_cells
        self._string_threads = string_threads                     # for the per-cell strings (combo, reasons) made right here
        self._late = [] if late_text else None                    # (ptr, len, seq, ascii) of strings allocated now and written by finish_text()
        self.status = _view(L.dyd_split_status(handle), np.uint8, n_cells).copy()
        self.n_expanded = _view(L.dyd_split_n_expanded(handle), np.int32, n_cells).copy()
        rows, events = int(L.dyd_split_rows(handle)), int(L.dyd_split_events(handle))
        self.n_records = rows
        self.row_cell = _view(L.dyd_split_row_cell(handle), np.int64, rows).copy()
        self.row_label = _view(L.dyd_split_row_label(handle), np.int32, rows).copy()
        self.event_cell = _view(L.dyd_split_event_cell(handle), np.int64, events).copy()
        self.event_kind = _view(L.dyd_split_event_kind(handle), np.uint8, events).copy()
        self.event_code = _view(L.dyd_split_event_code(handle), np.int32, events).copy()
        p, ln = C.c_void_p(), C.c_void_p()
        _native.check(L.dyd_split_rec_views(handle, C.byref(p), C.byref(ln)), "dyd_split_rec_views")
        self.rec_ptr = _view(p.value, np.uint64, rows)            # views into the handle
        self.rec_len = _view(ln.value, np.int64, rows)
        self.fast_cells = int(L.dyd_split_fast_cells(handle))
        self.all_ascii = bool(L.dyd_split_all_ascii(handle))
        secs = np.zeros(2, np.float64)
        L.dyd_split_seconds(handle, secs.ctypes.data)
        self.seconds = {"parse": float(secs[0]), "gather": float(secs[1])}
        self.combo = self._strings(1, n_cells)
        self.reasons = self._reason_strings(n_cells)
        self.reasons_nonempty = np.diff(self._buffers(2, n_cells)[1]) > 0 if n_cells else np.zeros(0, bool)
        self.undefined = self._strings(4, int(L.dyd_split_undefined(handle)))      # distinct labels event_code indexes
        self._n_labels = None

    def _buffers(self, which, count):
        d, o = C.c_void_p(), C.c_void_p()
        _native.check(_native.load_library().dyd_split_strings(self._h, which, C.byref(d), C.byref(o)), "dyd_split_strings")
        off = _view(o.value, np.int64, count + 1)
        return _view(d.value, np.uint8, int(off[-1]) if count else 0), off

    def _strings(self, which, count) -> np.ndarray:
        if count == 0:
            return np.empty(0, object)
        text, off = self._buffers(which, count)
        if self._late is not None:
            from . import pycells
            if pycells.available() and count >= pycells.MIN_THREADED // 16:    # allocate now (the GIL's part), write later on all cores
                ptr = (off[:-1] + (text.ctypes.data if len(text) else 0)).astype(np.uint64)
                length = np.diff(off)
                seq, asc = pycells.alloc_strings(ptr, length, n_threads=max(1, self._string_threads))   # (short texts: no helper threads)
                self._late.append((ptr, length, seq, asc))
                return seq
        return strings_from_buffers(text, off, n_threads=self._string_threads)

    def _reason_strings(self, count) -> np.ndarray:
        """per cell its joined reasons ("" for none): one str per DISTINCT text, shared by the cells that carry it — the texts name a
        row's undefined labels, so a table holds a handful of them (they are Chinese: a decode each otherwise)"""
        L = _native.load_library()
        codes_ptr = L.dyd_split_reason_code(self._h) if count else None
        if not codes_ptr:
            return self._strings(2, count)
        from . import pycells

        n_distinct = int(L.dyd_split_reason_distinct(self._h))
        codes = _view(codes_ptr, np.int32, count)
        first = _view(L.dyd_split_reason_first(self._h), np.int64, n_distinct)
        text, off = self._buffers(2, count)
        table = np.empty(n_distinct + 1, object)
        for k, cell in enumerate(first.tolist()):
            table[k] = bytes(text[off[cell]:off[cell + 1]]).decode("utf-8")
        table[n_distinct] = ""
        codes = np.where(codes < 0, n_distinct, codes).astype(np.int32)
        if pycells.available():
            return pycells.take_small(table, codes, n_threads=self._string_threads)
        return table[codes]

    def finish_text(self):
        """writes the text of the per-cell strings that were only allocated so far (``late_text``); the handle must still be open"""
        from . import pycells

        late, self._late = self._late, None
        for ptr, length, seq, asc in late or ():
            pycells.fill_strings(ptr, length, seq, asc)

    def label_stats(self, n_labels: int):
        """per label of the rules: (index of the first record carrying it or -1, number of records)"""
        L = _native.load_library()
        return (_view(L.dyd_split_label_first(self._h), np.int64, n_labels).copy(),
                _view(L.dyd_split_label_count(self._h), np.int64, n_labels).copy())

    def record_strings(self, idx=None, slot=None) -> np.ndarray:
        """object array of the records' JSON text (str): out[slot[i]] = record idx[i] (either optional; all records in row order
        without both)"""
        from . import pycells

        if self._h is None:
            raise ValueError("the expansion was closed")
        if pycells.available():
            return pycells.strings_from_views(self.rec_ptr, self.rec_len, idx, all_ascii=self.all_ascii, slot=slot, checked=slot is not None and idx is None)
        text, off = self._buffers(0, self.n_records)                # portable route: one flat copy, then pyarrow
        got = strings_from_buffers(text, off)
        if idx is not None:
            got = got[idx]
        if slot is None:
            return got
        out = np.empty(len(got), object)
        out[slot] = got
        return out

    def record_text(self, idx=None, slot=None):
        """(contiguous utf-8 buffer, offsets [n+1]) with record idx[i] at place slot[i] — for an Arrow string column"""
        from . import pycells

        if self._h is None:
            raise ValueError("the expansion was closed")
        if pycells.available():
            return pycells.gather_text(self.rec_ptr, self.rec_len, idx, slot=slot)
        text, off = self._buffers(0, self.n_records)
        order = np.arange(self.n_records, dtype=np.int64) if idx is None else np.asarray(idx, np.int64)
        if slot is not None:
            inv = np.empty(len(order), np.int64)
            inv[slot] = order
            order = inv
        lens = np.diff(off)[order]
        out_off = np.zeros(len(order) + 1, np.int64)
        np.cumsum(lens, out=out_off[1:])
        pieces = [text[off[k]:off[k + 1]] for k in order.tolist()]
        return (np.concatenate(pieces) if pieces else np.zeros(0, np.uint8)), out_off

    @property
    def row_json(self) -> np.ndarray:
        return self.record_strings()

    @property
    def event_label(self) -> np.ndarray:
        """per event the undefined label ("" for the other kinds)"""
        table = np.concatenate([self.undefined, np.asarray([""], object)])
        return table[self.event_code]                               # code -1 -> the trailing ""

    def close(self):
        if self._h:
            self.rec_ptr = self.rec_len = None
            _native.load_library().dyd_split_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def _label_buffers(labels):
    lab = [s.encode("utf-8") for s in labels]
    lab_off = np.zeros(len(lab) + 1, np.int64)
    np.cumsum([len(b) for b in lab], out=lab_off[1:])
    return np.frombuffer(b"".join(lab) or b"\0", dtype=np.uint8), lab_off


def split_expand(cells, labels: list, n_threads: int = 0) -> SplitExpansion:
    """cells: per row the JSON cell the reference would pick (str) or None; labels: keys of label_to_category"""
    L = _native.load_library()
    buf, off, missing, keep = cells_to_buffers(cells)
    for i, c in enumerate(cells):                       # "" is not a usable cell either (processor.py:716); in the joined buffer it
        if c == "":                                     # is followed by its separator blank, so say so here
            missing[i] = 1
    lab_buf, lab_off = _label_buffers(labels)
    h = C.c_void_p()
    _native.check(L.dyd_json_split_expand(buf.ctypes.data, off.ctypes.data, missing.ctypes.data, len(cells), lab_buf.ctypes.data,
                                          lab_off.ctypes.data, len(labels), n_threads, C.byref(h)), "dyd_json_split_expand")
    return SplitExpansion(h, len(cells))


def split_expand_views(ptr: np.ndarray, length: np.ndarray, missing: np.ndarray, labels: list, n_threads: int = 0) -> SplitExpansion:
    """the same over one (address, length) view per cell (pycells.CellViews of the DataFrame's own str objects)"""
    L = _native.load_library()
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.int64)
    missing = np.ascontiguousarray(missing, dtype=np.uint8)
    lab_buf, lab_off = _label_buffers(labels)
    h = C.c_void_p()
    _native.check(L.dyd_json_split_expand_v(ptr.ctypes.data, length.ctypes.data, missing.ctypes.data, len(ptr), lab_buf.ctypes.data,
                                            lab_off.ctypes.data, len(labels), n_threads, C.byref(h)), "dyd_json_split_expand_v")
    return SplitExpansion(h, len(ptr))


class SplitExpansionBatches:
    """The expansion of a large table made in row batches, with the interface of SplitExpansion.  While native threads parse
    batch b + 1, the calling thread — the one holding the GIL — allocates the str objects of batch b's records (lengths are final
    once a batch is parsed; pycells.alloc_strings).  The texts are written, and the strings moved to their places in the category
    frames, only when ``record_strings`` is told the places (pycells.fill_strings, worker threads, no GIL): the allocation, which
    is the serial part of building 14 M strings, is hidden behind the parse instead of following it."""

    def __init__(self, parts, bounds, shells):
        self._parts, self._bounds = parts, bounds
        self._shells = shells                                  # per batch (seq, ascii) or None
        self.n_cells = bounds[-1][1] if bounds else 0
        cat = np.concatenate
        self.status = cat([p.status for p in parts])
        self.n_expanded = cat([p.n_expanded for p in parts])
        self.combo = cat([p.combo for p in parts])
        self.reasons = cat([p.reasons for p in parts])
        self.reasons_nonempty = cat([p.reasons_nonempty for p in parts])
        self.row_cell = cat([p.row_cell + lo for p, (lo, _) in zip(parts, bounds)])
        self.row_label = cat([p.row_label for p in parts])
        self.event_cell = cat([p.event_cell + lo for p, (lo, _) in zip(parts, bounds)])
        self.event_kind = cat([p.event_kind for p in parts])
        undef, codes = {}, []
        for p in parts:                                        # the batches' tables of undefined labels become one
            remap = np.asarray([undef.setdefault(lab, len(undef)) for lab in p.undefined.tolist()] + [-1], np.int32)
            codes.append(remap[p.event_code])                  # code -1 -> the trailing -1
        self.event_code = cat(codes)
        self.undefined = np.empty(len(undef), object)
        self.undefined[:] = list(undef)
        self.n_records = int(sum(p.n_records for p in parts))
        self._rec_base = np.zeros(len(parts) + 1, np.int64)
        np.cumsum([p.n_records for p in parts], out=self._rec_base[1:])
        self.rec_ptr = cat([p.rec_ptr for p in parts])         # the handles stay open until close()
        self.rec_len = cat([p.rec_len for p in parts])
        self.fast_cells = int(sum(p.fast_cells for p in parts))
        self.all_ascii = all(p.all_ascii for p in parts)
        self.seconds = {k: float(sum(p.seconds[k] for p in parts)) for k in ("parse", "gather")}
        self._open = True

    def label_stats(self, n_labels: int):
        first = np.full(n_labels, -1, np.int64)
        count = np.zeros(n_labels, np.int64)
        for p, base in zip(self._parts, self._rec_base[:-1].tolist()):
            f, c = p.label_stats(n_labels)
            take = (f >= 0) & (first < 0)
            first[take] = f[take] + base
            count += c
        return first, count

    def record_strings(self, idx=None, slot=None) -> np.ndarray:
        from . import pycells

        if not self._open:
            raise ValueError("the expansion was closed")
        shells, self._shells = self._shells, None              # usable once: the strings move out
        if shells is None or idx is not None:
            shells = None
            return pycells.strings_from_views(self.rec_ptr, self.rec_len, idx, all_ascii=self.all_ascii, slot=slot,
                                              checked=slot is not None and idx is None)
        if slot is None:
            return np.concatenate([pycells.fill_strings(p.rec_ptr, p.rec_len, seq, asc) for p, (seq, asc) in zip(self._parts, shells)])
        slot = np.ascontiguousarray(slot, dtype=np.int64)
        if len(slot) != self.n_records:
            raise ValueError("record_strings: one slot per record")
        out = np.empty(self.n_records, object)
        for p, (seq, asc), lo, hi in zip(self._parts, shells, self._rec_base[:-1].tolist(), self._rec_base[1:].tolist()):
            pycells.fill_strings(p.rec_ptr, p.rec_len, seq, asc, slot[lo:hi], out)
        return out

    def record_text(self, idx=None, slot=None):
        from . import pycells

        if not self._open:
            raise ValueError("the expansion was closed")
        return pycells.gather_text(self.rec_ptr, self.rec_len, idx, slot=slot)

    @property
    def row_json(self) -> np.ndarray:
        return self.record_strings()

    def close(self):
        if self._open:
            self._open = False
            self._shells = None                                # strings never handed out are released unwritten
            self.rec_ptr = self.rec_len = None
            for p in self._parts:
                p.close()

    def __del__(self):
        self.close()


def split_expand_views_batched(ptr: np.ndarray, length: np.ndarray, missing: np.ndarray, labels: list, n_batches: int = 6,
                               allocate: bool = True) -> SplitExpansionBatches:
    """split_expand_views over ``n_batches`` row ranges, one native call each on a helper thread; the calling thread turns every
    finished batch into its arrays and (``allocate``) the still-empty str objects of its records meanwhile."""
    import queue
    import threading

    from . import pycells

    L = _native.load_library()
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.int64)
    missing = np.ascontiguousarray(missing, dtype=np.uint8)
    n = len(ptr)
    n_batches = max(1, min(int(n_batches), n))
    bounds = [(n * b // n_batches, n * (b + 1) // n_batches) for b in range(n_batches)]
    lab_buf, lab_off = _label_buffers(labels)
    threads = max(1, len(os.sched_getaffinity(0)) - 1) if hasattr(os, "sched_getaffinity") else 0
    ready = queue.Queue()
    stop = threading.Event()

    def parse_all():
        for lo, hi in bounds:
            if stop.is_set():
                break
            h = C.c_void_p()
            rc = L.dyd_json_split_expand_v(ptr[lo:hi].ctypes.data, length[lo:hi].ctypes.data, missing[lo:hi].ctypes.data, hi - lo,
                                           lab_buf.ctypes.data, lab_off.ctypes.data, len(labels), threads, C.byref(h))
            ready.put((rc, h))
            if rc != 0:
                break

    worker = threading.Thread(target=parse_all, name="dyd-split-parse", daemon=True)
    worker.start()
    parts, shells = [], []
    allocate = allocate and pycells.available()
    try:
        for lo, hi in bounds:
            rc, h = ready.get()
            _native.check(rc, "dyd_json_split_expand_v")
            part = SplitExpansion(h, hi - lo, string_threads=1, late_text=allocate)   # (the cores are parsing the next batch)
            parts.append(part)
            if allocate:
                shells.append(pycells.alloc_strings(part.rec_ptr, part.rec_len, part.all_ascii))
    except BaseException:
        stop.set()
        worker.join()
        while not ready.empty():
            rc, h = ready.get()
            if rc == 0 and h:
                L.dyd_split_free(h)
        shells = None
        for part in parts:
            part.close()
        raise
    worker.join()
    for part in parts:
        part.finish_text()
    return SplitExpansionBatches(parts, bounds, shells if allocate else None)


# ------------------------------------------------------------------------------------------ label_replace step
RL_REWRITTEN, RL_EMPTY, RL_UNDECODABLE, RL_UNCHANGED, RL_IRREGULAR = 0, 1, 2, 3, 5


def _packed(strings):
    raw = [s.encode("utf-8") for s in strings]
    off = np.zeros(len(raw) + 1, np.int64)
    np.cumsum(np.fromiter(map(len, raw), dtype=np.int64, count=len(raw)), out=off[1:])
    return np.frombuffer(b"".join(raw) or b"\0", dtype=np.uint8), off


class Relabelling:
    """Result of relabel: per cell status, counts[n, 5] (objects, names missing, labels, labels replaced, objects
    renamed), has_diff, the unmatched labels in order of appearance (token, token_cell); the text after the step (the
    cell itself unless it was rewritten, empty for skipped cells) and the joined old / new names as flat buffers
    (text_data / text_off ...) or, through text() / before() / after(), as object arrays of str."""

    def __init__(self, handle, n_cells, keep=None):
        L = _native.load_library()
        self._h, self._keep, self.n_cells = handle, keep, n_cells
        self.status = _view(L.dyd_relabel_status(handle), np.uint8, n_cells).copy()
        self.has_diff = _view(L.dyd_relabel_has_diff(handle), np.uint8, n_cells).copy()
        self.counts = _view(L.dyd_relabel_counts(handle), np.int32, 5 * n_cells).reshape(-1, 5).copy()
        tokens = int(L.dyd_relabel_tokens(handle))
        self.token_cell = _view(L.dyd_relabel_token_cell(handle), np.int64, tokens).copy()
        self.token = self._strings(3, tokens)

    def _buffers(self, which, count):
        d, o = C.c_void_p(), C.c_void_p()
        _native.check(_native.load_library().dyd_relabel_strings(self._h, which, C.byref(d), C.byref(o)), "dyd_relabel_strings")
        off = _view(o.value, np.int64, count + 1)
        return _view(d.value, np.uint8, max(int(off[-1]), 1) if count else 0), off, d.value, o.value

    def _strings(self, which, count):
        _, _, d, o = self._buffers(which, count)
        return _strings(d, o, count)

    def text(self):
        return self._strings(0, self.n_cells)

    def before(self):
        return self._strings(1, self.n_cells)

    def after(self):
        return self._strings(2, self.n_cells)

    def text_buffers(self):
        """(data u8, off i64) views into the handle: valid until close()"""
        data, off, _, _ = self._buffers(0, self.n_cells)
        return data, off

    def close(self):
        if self._h is not None:
            _native.load_library().dyd_relabel_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def relabel_buffers(data, off, missing, label_map: dict, n_threads: int = 0, keep=None) -> Relabelling:
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    missing = np.ascontiguousarray(missing, dtype=np.uint8)
    L = _native.load_library()
    kbuf, koff = _packed(list(label_map.keys()))
    vbuf, voff = _packed(list(label_map.values()))
    h = C.c_void_p()
    _native.check(L.dyd_json_relabel(data.ctypes.data, off.ctypes.data, missing.ctypes.data, len(off) - 1, kbuf.ctypes.data, koff.ctypes.data,
                                     vbuf.ctypes.data, voff.ctypes.data, len(label_map), n_threads, C.byref(h)), "dyd_json_relabel")
    return Relabelling(h, len(off) - 1, (keep, data, off, missing))


def relabel(cells, label_map: dict, n_threads: int = 0) -> Relabelling:
    """cells: JSON cells (str) or anything else for a cell the step skips; label_map: old label -> new label"""
    buf, off, missing, keep = cells_to_buffers(cells)
    for i, c in enumerate(cells):                       # "" is skipped too (reference processor.py:571)
        if c == "":
            missing[i] = 1
    return relabel_buffers(buf, off, missing, label_map, n_threads, keep)


# ------------------------------------------------------------------------------------------ YOLO step
class LabelledScan(_Scan):
    """labelled boxes per cell: box4 (min x, min y, max x, max y), cell_box_off, sel (name == the row's label)"""

    def __init__(self, handle, n_cells, keep):
        super().__init__(handle, n_cells, keep)
        L = _native.load_library()
        nb = int(self.cell_box_off[-1]) if n_cells else 0
        self.n_boxes = nb
        self.box4 = _view(L.dyd_scan_xy(handle), np.float64, 4 * nb).reshape(-1, 4)
        self.sel = _view(L.dyd_scan_sel(handle), np.uint8, nb)


def scan_labelled(cells, labels, n_threads: int = 0) -> LabelledScan:
    """cells: annotation JSON per row; labels: the row's label value (str) per row"""
    L = _native.load_library()
    buf, off, missing, keep = cells_to_buffers(cells)
    lab = [s.encode("utf-8") for s in labels]
    lab_off = np.zeros(len(lab) + 1, np.int64)
    np.cumsum(np.fromiter(map(len, lab), dtype=np.int64, count=len(lab)), out=lab_off[1:])
    lab_buf = np.frombuffer(b"".join(lab) or b"\0", dtype=np.uint8)
    h = C.c_void_p()
    _native.check(L.dyd_json_scan_labelled(buf.ctypes.data, off.ctypes.data, missing.ctypes.data, len(cells), lab_buf.ctypes.data,
                                           lab_off.ctypes.data, n_threads, C.byref(h)), "dyd_json_scan_labelled")
    return LabelledScan(h, len(cells), (keep, lab_buf, lab_off))
