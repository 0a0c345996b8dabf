"""Native CSV hand-off (csrc/host_csv.cpp).

``read_split`` reads a step's input CSV so that the HEAVY columns (the two annotation JSON columns)
never become Python objects: they are returned as flat utf-8 buffers for the native JSON scanner and
for ``write_table``; every other (light) column is parsed by pandas itself from a reduced CSV text, so
dtype inference, NA handling and float parsing stay exactly pandas'.

``write_table`` writes typed column buffers the way ``DataFrame.to_csv(index=False)`` does
(csv.QUOTE_MINIMAL, float repr, NaN -> empty) and checks a sample of rows against pandas before it
trusts the native writer for the whole file.

Both return ``None`` / ``False`` whenever the file or the table is outside what the fast path
reproduces exactly; the callers then take the plain pandas path.
"""
from __future__ import annotations

import csv
import ctypes as C
import io
import os
from dataclasses import dataclass

import numpy as np
import pandas as pd

from . import _native

_BOM = b"\xef\xbb\xbf"


def enabled() -> bool:
    return os.environ.get("DYD_NATIVE_CSV", "1") != "0"


def _quote_cr() -> bool:
    """does this interpreter's csv module quote a field that holds a bare CR?  (changed across versions)"""
    s = io.StringIO()
    csv.writer(s, lineterminator="\n").writerow(["a\rb", "c"])
    return s.getvalue().startswith('"')


_QUOTE_CR = _quote_cr()


@dataclass
class Utf8Column:
    """one string column as flat bytes: cell i = data[off[i]:off[i+1]], na[i] != 0 -> missing"""
    data: np.ndarray
    off: np.ndarray
    na: np.ndarray
    keep: object = None        # keeps foreign memory alive

    def __len__(self):
        return len(self.off) - 1

    def cell(self, i: int):
        if self.na[i]:
            return np.nan
        return bytes(self.data[self.off[i]:self.off[i + 1]]).decode("utf-8")

    def cells(self, rows) -> list:
        return [self.cell(int(i)) for i in rows]


def _view(ptr, dtype, count):
    if count == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))


@dataclass
class SplitTable:
    names: list                 # all column names in file order
    n_rows: int
    light: pd.DataFrame         # the non-heavy columns, parsed by pandas (RangeIndex)
    heavy: dict                 # name -> Utf8Column


def _utf8_like(encoding: str) -> bool:
    return encoding.lower().replace("_", "-") in ("utf-8-sig", "utf-8", "utf8")


class CsvIndex:
    """A tokenised CSV buffer (dyd_csv handle).  ``open`` returns None when the native tokeniser does not
    reproduce pandas on this text (CR line ends, ragged rows, stray quotes, duplicate / empty header names)."""

    def __init__(self, handle, buf, names, n_rows):
        self._h, self._buf, self.names, self.n_rows = handle, buf, names, n_rows

    @classmethod
    def open(cls, buf: np.ndarray):
        L = _native.load_library()
        if buf.size == 0:
            return None
        h = C.c_void_p()
        if L.dyd_csv_index(buf.ctypes.data, buf.size, C.byref(h)) != 0:
            return None
        n_rows, n_cols = int(L.dyd_csv_rows(h)), int(L.dyd_csv_cols(h))
        names, tmp = [], np.empty(4096, np.uint8)
        for c in range(n_cols):
            n = L.dyd_csv_header(h, c, tmp.ctypes.data, tmp.size)
            if n < 0:
                L.dyd_csv_free(h)
                return None
            names.append(bytes(tmp[:n]).decode("utf-8"))
        if len(set(names)) != len(names) or any(nm == "" or nm.startswith("Unnamed") for nm in names):
            L.dyd_csv_free(h)                                # pandas renames such columns: leave it to pandas
            return None
        return cls(h, buf, names, n_rows)

    def close(self):
        if self._h is not None:
            try:
                _native.load_library().dyd_csv_free(self._h)
            except Exception:  # noqa: BLE001 - interpreter shutdown: the library may already be gone
                pass
            self._h = None

    def __del__(self):
        # dropping the index of a large file unmaps gigabytes (0.13 s per 1.6 GB file): nobody waits for that — a thread of its own
        h = self._h
        if h is not None and self._buf is not None and self._buf.size >= (64 << 20):
            import sys
            import threading
            try:
                if not sys.is_finalizing():
                    free = _native.load_library().dyd_csv_free
                    self._h = None
                    threading.Thread(target=free, args=(h,), name="dyd-csv-free", daemon=True).start()
                    return
            except Exception:  # noqa: BLE001 - no thread to be had: free it here
                self._h = h
        self.close()

    def has_cr(self) -> bool:
        """some lines of the file end with CR LF (a text-mode reader would rewrite those; path readers do not)"""
        return bool(_native.load_library().dyd_csv_has_cr(self._h))

    def col_bytes(self, c: int) -> int:
        return int(_native.load_library().dyd_csv_col_bytes(self._h, c))

    def row_end(self, row: int) -> int:
        """byte offset (inside the indexed buffer) just behind data row `row`; -1 = the header line"""
        return int(_native.load_library().dyd_csv_row_end(self._h, row))

    def extract(self, c: int):
        """column c as Utf8Column, or None when pandas might not type it as str in every piece of the file"""
        L = _native.load_library()
        pb, po, pn = C.c_void_p(), C.c_void_p(), C.c_void_p()
        if L.dyd_csv_extract(self._h, c, C.byref(pb), C.byref(po), C.byref(pn)) != 0:
            return None
        off = _view(po.value, np.int64, self.n_rows + 1)          # views into the handle: the column keeps it alive
        data = _view(pb.value, np.uint8, int(off[-1]) + 1)
        na = _view(pn.value, np.uint8, self.n_rows)
        # pandas infers dtypes per low-memory piece of the file, and a piece whose present cells all look
        # numeric / boolean becomes numbers ("1.50" -> 1.5).  The column is taken natively only when NO
        # present cell could be read as a number / boolean (na == 2): then every piece is object / str.
        if (na == 2).any() or (na != 0).all():
            return None
        return Utf8Column(data, off, na, keep=self)

    def project(self, keep: list):
        """CSV text of the file's width in which only the columns `keep` carry their cells (for pandas)"""
        L = _native.load_library()
        arr = np.asarray(keep, np.int32)
        pt, ln = C.c_void_p(), C.c_int64()
        if L.dyd_csv_project(self._h, arr.ctypes.data, len(arr), C.byref(pt), C.byref(ln)) != 0:
            return None
        return bytes(_view(pt.value, np.uint8, ln.value))


def read_split(path: str, heavy_names, encoding: str = "utf-8-sig"):
    """-> SplitTable, or None when the fast path must not be used for this file."""
    if not enabled() or not _utf8_like(encoding):
        return None
    # the file is mapped, not read: the tokeniser's threads fault the page-cache pages in side by side (a single f.read() of a
    # 1.5 GB file took more than tokenising, extracting and parsing it together)
    import mmap
    with open(path, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        raw = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if size else b""
    start = len(_BOM) if raw[:3] == _BOM and "sig" in encoding.lower() else 0
    idx = CsvIndex.open(np.frombuffer(raw, dtype=np.uint8)[start:])
    if idx is None:
        return None
    names, n_rows = idx.names, idx.n_rows
    heavy = {}
    for nm in heavy_names:
        if nm not in names:
            continue
        col = idx.extract(names.index(nm))
        if col is None:
            return None
        heavy[nm] = col
    light_idx = [i for i, nm in enumerate(names) if nm not in heavy]
    if light_idx:
        text = idx.project(light_idx)
        if text is None:
            return None
        light_names = [names[i] for i in light_idx]
        light = pd.read_csv(io.BytesIO(text), encoding="utf-8", usecols=light_names)[light_names]
        if len(light) != n_rows or list(light.columns) != light_names:
            return None
    else:
        light = pd.DataFrame(index=pd.RangeIndex(n_rows))
    if not heavy:
        idx.close()
    return SplitTable(names, n_rows, light, heavy)


def frame_from_split(table: "SplitTable", rows=None) -> pd.DataFrame:
    """The DataFrame pandas.read_csv would have produced (for `rows`, index reset when rows is given):
    heavy columns are materialised as Python str / NaN only here, for callers that must return a frame."""
    import pyarrow as pa

    idx = np.arange(table.n_rows) if rows is None else np.asarray(rows, np.int64)
    out = {}
    for nm in table.names:
        if nm in table.heavy:
            col = table.heavy[nm]
            arr = pa.LargeStringArray.from_buffers(table.n_rows, pa.py_buffer(col.off), pa.py_buffer(col.data))
            vals = arr.take(pa.array(idx)).to_numpy(zero_copy_only=False).astype(object)
            vals[col.na[idx] != 0] = np.nan
            out[nm] = vals
        else:
            out[nm] = table.light[nm].to_numpy()[idx] if rows is not None else table.light[nm].to_numpy()
    df = pd.DataFrame(out, columns=table.names)
    for nm in table.light.columns:                      # keep pandas' dtypes (object columns stay object)
        if df[nm].dtype != table.light[nm].dtype:
            df[nm] = df[nm].astype(table.light[nm].dtype)
    return df


# ------------------------------------------------------------------------------------------------- writer
def _series_column(s: pd.Series):
    """pandas column -> (kind, arrays...) or None if the dtype is not covered"""
    kind = s.dtype.kind
    if s.dtype == np.int64:
        return (1, np.ascontiguousarray(s.to_numpy()), None, None)
    if s.dtype == np.float64:
        return (2, np.ascontiguousarray(s.to_numpy()), None, None)
    if s.dtype == np.bool_:
        return (3, np.ascontiguousarray(s.to_numpy().astype(np.uint8)), None, None)
    if kind == "O":
        from . import pycells
        if pycells.available() and len(s) >= 4096:
            # a column of str cells (missing ones aside) is copied out of the str objects' own UTF-8 by worker threads: no per-cell
            # Python.  Missing = what to_csv prints as the empty field (None, NaN, NaT, pd.NA).
            arr = s.to_numpy()
            na_b = np.zeros(len(arr), bool) if pycells.all_str(arr) else np.asarray(pd.isna(arr), dtype=bool)
            try:
                flat = pycells.flat_utf8(arr, na_b)
            except UnicodeEncodeError:
                flat = None
            if flat is not None:                            # (None: some present cell is not a str -> the per-cell walk below)
                return (0, flat[0] if len(flat[0]) else np.zeros(1, np.uint8), flat[1], na_b.astype(np.uint8))
        vals = s.tolist()
        na = np.fromiter((v is None or (type(v) is float and v != v) for v in vals), dtype=np.uint8, count=len(vals))
        # csv.writer prints str(value) for the scalars an object column can hold after JSON / CSV parsing
        ok = (str, int, float, bool, np.integer, np.floating, np.bool_)
        if not all(n or isinstance(v, ok) for v, n in zip(vals, na.tolist())):
            return None
        texts = [("" if n else (v if type(v) is str else str(v))) for v, n in zip(vals, na.tolist())]
        blob = "".join(texts)
        data = blob.encode("utf-8")
        lens = np.fromiter(map(len, texts), dtype=np.int64, count=len(texts))
        if len(data) != len(blob):
            for i, v in enumerate(texts):
                if not v.isascii():
                    lens[i] = len(v.encode("utf-8"))
        off = np.zeros(len(texts) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return (0, np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, np.uint8), off, na)
    return None


class _Cols(C.Structure):
    _fields_ = [("kind", C.c_int32), ("data", C.c_void_p), ("off", C.c_void_p), ("na", C.c_void_p)]


def write_table(path: str, names: list, columns: list, n_rows: int, rows=None, encoding: str = "utf-8-sig",
                check_rows: int = 40, append: bool = False, header: bool = True) -> bool:
    """Write the table like ``DataFrame(...)[names].to_csv(path, index=False, encoding=encoding)``.

    columns[i] is a pandas Series (light column, length n_rows) or a Utf8Column.  ``rows`` selects and
    orders source rows (default: all).  Returns False — without touching ``path`` — when a column type is
    not covered or when the sample check against pandas disagrees.  ``append`` / ``header`` = to_csv's
    mode="a" / header= (an appended part carries no BOM, like a text file opened for append at a non-zero offset)."""
    job = _prepare_write(path, names, columns, n_rows, rows, encoding, check_rows, append, header)
    return job is not None and job()


def write_tables(tables: list, encoding: str = "utf-8-sig") -> bool:
    """Several files at once: tables[i] = (path, names, columns, n_rows, rows).  Every table is checked against pandas first —
    False, with nothing written, if any of them is refused — and then the files are written side by side, one thread each on top
    of the writer's own: a buffered write holds its file's inode lock (tmpfs, ext4 and xfs alike), so the sixteen pwrite streams of
    ONE file take turns, while different files proceed in parallel."""
    import threading

    jobs = [_prepare_write(path, names, columns, n_rows, rows, encoding) for path, names, columns, n_rows, rows in tables]
    if any(j is None for j in jobs):
        return False
    if len(jobs) == 1:
        return jobs[0]()
    ok = [False] * len(jobs)

    def run(i):
        ok[i] = jobs[i]()

    threads = [threading.Thread(target=run, args=(i,), name="dyd-csv-write") for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return all(ok)


class WriteInBackground:
    """one prepared table being written by a thread of its own (the writer's sixteen on top): ``done()`` joins -> True on success"""

    def __init__(self, job):
        import threading

        self._ok = False
        self._thread = threading.Thread(target=self._run, args=(job,), name="dyd-csv-write")
        self._thread.start()

    def _run(self, job):
        self._ok = bool(job())

    def done(self) -> bool:
        self._thread.join()
        return self._ok


def prepare_write(path: str, names: list, columns: list, n_rows: int, rows=None, encoding: str = "utf-8-sig"):
    """write_table in two halves: the column buffers and the sample check against pandas now -> a callable that writes the file
    (True on success), or None when the table is refused (nothing written)"""
    return _prepare_write(path, names, columns, n_rows, rows, encoding)


def _prepare_write(path: str, names: list, columns: list, n_rows: int, rows=None, encoding: str = "utf-8-sig",
                   check_rows: int = 40, append: bool = False, header: bool = True):
    """everything of write_table up to the file itself: column buffers, the sample check against pandas.  -> a callable that
    writes the file (True on success), or None when the table is outside what the native writer reproduces."""
    if not enabled() or not _utf8_like(encoding):
        return None
    L = _native.load_library()
    specs, keep = [], []
    for col in columns:
        if isinstance(col, Utf8Column):
            specs.append((0, col.data, col.off, col.na))
        else:
            sp = _series_column(col)
            if sp is None:
                return None
            specs.append(sp)
    arr = (_Cols * len(specs))()
    for i, (kind, data, off, na) in enumerate(specs):
        data = np.ascontiguousarray(data)
        keep.append(data)
        arr[i].kind = kind
        arr[i].data = data.ctypes.data
        if off is not None:
            off = np.ascontiguousarray(off, dtype=np.int64); keep.append(off); arr[i].off = off.ctypes.data
        if na is not None:
            na = np.ascontiguousarray(na, dtype=np.uint8); keep.append(na); arr[i].na = na.ctypes.data
    header_line = pd.DataFrame(columns=names).to_csv(index=False).encode("utf-8")
    rows_arr = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
    n_out = n_rows if rows_arr is None else len(rows_arr)

    # ---- cross-check a sample of rows against pandas itself ---------------------------------------
    if n_out:
        pick = np.unique(np.concatenate([np.arange(min(check_rows, n_out)), np.arange(max(0, n_out - 8), n_out)]))
        src = pick if rows_arr is None else rows_arr[pick]
        sample = {}
        for nm, col in zip(names, columns):
            sample[nm] = col.cells(src) if isinstance(col, Utf8Column) else col.iloc[src].reset_index(drop=True)
        want = pd.DataFrame(sample, columns=names).to_csv(index=False).encode("utf-8")
        mem, ln = C.c_void_p(), C.c_int64()
        srcc = np.ascontiguousarray(src, dtype=np.int64)
        rc = L.dyd_csv_write(None, header_line, len(header_line), arr, len(specs), n_rows, srcc.ctypes.data, len(srcc), int(_QUOTE_CR), 1,
                             1, C.byref(mem), C.byref(ln))
        if rc != 0:
            return None
        got = bytes(_view(mem.value, np.uint8, ln.value))
        L.dyd_host_free(mem)
        if got != want:
            return None

    buffers = (keep, specs, rows_arr)          # what `arr` points into: bound to the job below, alive as long as it is

    def write(_buffers=buffers) -> bool:
        bom = _BOM if "sig" in encoding.lower() and not (append and os.path.exists(path) and os.path.getsize(path) > 0) else b""
        full_header = bom + (header_line if header else b"")
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        rc = L.dyd_csv_write(os.fsencode(path), full_header, len(full_header), arr, len(specs), n_rows,
                             rows_arr.ctypes.data if rows_arr is not None else None, n_out, int(_QUOTE_CR), 0, 2 if append else 0,
                             None, None)
        return rc == 0

    return write
