"""Flatten: DataFrame cells -> the SoA buffers the device stage consumes, and back (emit).

Host-side work of the drop-in steps.  The structure walk below follows the reference's own
accessors step by step (``doc.get("objects", [])``, ``obj.get("polygon", {}).get("ptList", [])``,
the ``"x" in p and "y" in p`` guard ...) so that every malformed structure raises the same
exception type, at the same row, as ``core/processor.py:262-281`` / ``:341-366`` do.

Only coordinates that an IEEE double represents exactly go to the device: floats, bools and
ints of magnitude <= 2**53 (K1) / <= 2**25 (K2, where products of differences must also stay
exact).  Anything else in a coordinate slot (None, str, list, huge int) cannot be laid out as
f64 and is resolved right here with the same builtin ``min``/``max``/arithmetic the reference
applies to it, which is also what makes it raise TypeError exactly when the reference does.
``stats`` counts such boxes/rows so callers and tests can assert the device did the work.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np
import pandas as pd

_K1_INT_LIMIT = 2 ** 53
_K2_INT_LIMIT = 2 ** 25


def _f64_or_none(v, int_limit):
    t = type(v)
    if t is float:
        return v
    if t is int:
        return float(v) if -int_limit <= v <= int_limit else None
    if t is bool:
        return 1.0 if v else 0.0
    return None


# ------------------------------------------------------------------------------------------
# a3  polygon ptList -> bbox  (reference core/processor.py:262-281)
# ------------------------------------------------------------------------------------------
@dataclass
class PolygonBatch:
    docs: list                     # per cell: parsed document, or None when the cell yields None
    objs: list                     # per cell: list of the dict objects of the document (in order)
    box_pts: list                  # per box: its valid points (dicts) — the tokens emit re-uses
    host_boxes: dict               # box index -> ptList computed on the host (non-f64 coordinates)
    xy: np.ndarray                 # [P,2] f64
    pt_off: np.ndarray             # [B+1] i32
    cell_box_off: np.ndarray       # [n_cells+1] i32 first box of each cell
    stats: dict = field(default_factory=dict)


def _host_bbox(points):
    """get_bbox_points (:254-260) on the host for coordinates that are not f64-representable."""
    lo_x = min(p["x"] for p in points)
    hi_x = max(p["x"] for p in points)
    lo_y = min(p["y"] for p in points)
    hi_y = max(p["y"] for p in points)
    return [{"x": lo_x, "y": lo_y}, {"x": hi_x, "y": hi_y}]


def flatten_polygons(cells) -> PolygonBatch:
    docs, objs_per_cell, box_pts, host_boxes = [], [], [], {}
    xs, npts, cell_boxes = [], [], [0]
    n_host = 0
    for cell in cells:
        doc, dict_objs = None, []
        if isinstance(cell, str):                     # :264 — NaN / non-str cells give None
            try:
                doc = json.loads(cell)
            except json.JSONDecodeError:              # :280-281
                doc = None
            else:
                for obj in doc.get("objects", []):    # AttributeError / TypeError propagate like the reference
                    if not isinstance(obj, dict):     # :270 non-dict objects are dropped
                        continue
                    pts = [p for p in obj.get("polygon", {}).get("ptList", [])
                           if isinstance(p, dict) and "x" in p and "y" in p]   # :253
                    flat, ok = [], True
                    for p in pts:
                        x = _f64_or_none(p["x"], _K1_INT_LIMIT)
                        y = _f64_or_none(p["y"], _K1_INT_LIMIT)
                        if x is None or y is None:
                            ok = False
                            break
                        flat.append(x)
                        flat.append(y)
                    if not ok:                         # resolve with builtin min/max, raising as CPython does
                        host_boxes[len(box_pts)] = _host_bbox(pts)
                        n_host += 1
                        flat = []
                    xs.extend(flat)
                    npts.append(len(flat) // 2)
                    box_pts.append(pts)
                    dict_objs.append(obj)
        docs.append(doc)
        objs_per_cell.append(dict_objs)
        cell_boxes.append(len(box_pts))
    pt_off = np.zeros(len(npts) + 1, np.int64)
    np.cumsum(np.asarray(npts, np.int64), out=pt_off[1:])
    if pt_off[-1] >= 2 ** 31:
        raise OverflowError("more than 2^31 points in one batch: split the table into chunks")
    xy = np.asarray(xs, np.float64).reshape(-1, 2)
    return PolygonBatch(docs, objs_per_cell, box_pts, host_boxes, xy, pt_off.astype(np.int32),
                        np.asarray(cell_boxes, np.int32),
                        {"cells": len(docs), "boxes": len(box_pts), "points": int(pt_off[-1]),
                         "host_boxes": n_host})


def emit_polygons(batch: PolygonBatch, arg4: np.ndarray) -> list:
    """Re-serialise every document with each ptList replaced by its two corner points.

    ``arg4[b] = (argmin_x, argmin_y, argmax_x, argmax_y)`` from K1; the emitted coordinate is
    the ORIGINAL Python object at that index (so ``10`` stays ``10`` and ``10.0`` stays ``10.0``,
    processor.py:256-260), -1 marks a box without valid points (:254-255)."""
    out = []
    arg = np.asarray(arg4).tolist()
    for ci, doc in enumerate(batch.docs):
        if doc is None:
            out.append(None)
            continue
        b = int(batch.cell_box_off[ci])
        rewritten = []
        for obj in batch.objs[ci]:
            if b in batch.host_boxes:
                corners = batch.host_boxes[b]
            else:
                pts = batch.box_pts[b]
                a = arg[b]
                if not pts:
                    corners = [{"x": None, "y": None}, {"x": None, "y": None}]
                else:
                    corners = [{"x": pts[a[0]]["x"], "y": pts[a[1]]["y"]},
                               {"x": pts[a[2]]["x"], "y": pts[a[3]]["y"]}]
            new_obj = obj.copy()                      # :271
            if "polygon" not in new_obj:              # :274-275
                new_obj["polygon"] = {}
            new_obj["polygon"]["ptList"] = corners    # :276
            rewritten.append(new_obj)
            b += 1
        doc["objects"] = rewritten                    # :278
        out.append(json.dumps(doc, ensure_ascii=False))   # :279
    return out


# ------------------------------------------------------------------------------------------
# a4  two-point boxes per image row  (reference core/processor.py:341-366)
# ------------------------------------------------------------------------------------------
@dataclass
class BoxBatch:
    box4: np.ndarray               # [B,4] f64 (p1x,p1y,p2x,p2y) as stored; K2 normalises corners
    row_off: np.ndarray            # [n_rows+1] i32
    host_rows: dict                # row -> list of (x1,y1,x2,y2) Python tuples for non-f64 rows
    stats: dict = field(default_factory=dict)


def flatten_boxes(cells) -> BoxBatch:
    flat, counts, host_rows = [], [], {}
    for ri, cell in enumerate(cells):
        row_vals, tuples, device_ok = [], [], True
        try:                                           # :343 — any exception keeps the prefix
            if isinstance(cell, str):                  # :344
                doc = json.loads(cell)
                for obj in doc.get("objects", []):
                    if not isinstance(obj, dict):
                        continue
                    pts = obj.get("polygon", {}).get("ptList", [])
                    if len(pts) != 2:
                        continue
                    a, b = pts
                    if not (isinstance(a, dict) and isinstance(b, dict) and "x" in a and "y" in a
                            and "x" in b and "y" in b):
                        continue
                    raw = (a["x"], a["y"], b["x"], b["y"])
                    # :359-362 — evaluated here only to raise (None vs number ...) exactly where the
                    # reference's min()/max() would; the device redoes the normalisation in f64
                    tuples.append((min(raw[0], raw[2]), min(raw[1], raw[3]),
                                   max(raw[0], raw[2]), max(raw[1], raw[3])))
                    vals = [_f64_or_none(v, _K2_INT_LIMIT) for v in raw]
                    if device_ok and None not in vals:
                        row_vals.extend(vals)
                    else:
                        device_ok = False
        except Exception:                               # noqa: BLE001 (:364-365)
            pass
        if device_ok:
            flat.extend(row_vals)
            counts.append(len(row_vals) // 4)
        else:
            host_rows[ri] = tuples
            counts.append(0)
    row_off = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(np.asarray(counts, np.int64), out=row_off[1:])
    if row_off[-1] >= 2 ** 31:
        raise OverflowError("more than 2^31 boxes in one batch: split the table into chunks")
    return BoxBatch(np.asarray(flat, np.float64).reshape(-1, 4), row_off.astype(np.int32), host_rows,
                    {"rows": len(counts), "boxes": int(row_off[-1]), "host_rows": len(host_rows)})


def host_row_is_high(boxes, min_boxes, thr) -> bool:
    """meet_conditions / calculate_iou (:328-339, :368-376) for rows whose coordinates are not
    f64-representable (huge ints, strings): CPython arithmetic, raising as the reference does."""
    if len(boxes) < min_boxes:
        return False
    for i in range(len(boxes)):
        p = boxes[i]
        for j in range(i + 1, len(boxes)):
            q = boxes[j]
            inter = (max(0, min(p[2], q[2]) - max(p[0], q[0]))
                     * max(0, min(p[3], q[3]) - max(p[1], q[1])))
            if inter == 0:
                iou = 0.0
            else:
                union = (p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter
                iou = inter / union if union != 0 else 0.0
            if iou >= thr:
                return True
    return False


# ------------------------------------------------------------------------------------------
# a1 / a2  key column -> flat bytes + offsets for K3
# ------------------------------------------------------------------------------------------
NA_KEY = np.array([0x6e616e5f6b65795f, 0x5f5f6e615f5f6b79], np.uint64)   # key given to missing cells


def _strings_to_bytes(values) -> tuple:
    """list of str -> (uint8 buffer, int64 offsets)."""
    try:
        import pyarrow as pa
        arr = pa.array(values, type=pa.large_string())
        bufs = arr.buffers()
        off = np.frombuffer(bufs[1], dtype=np.int64, count=len(arr) + 1)
        data = np.frombuffer(bufs[2], dtype=np.uint8, count=int(off[-1])) if bufs[2] is not None else np.zeros(0, np.uint8)
        return data, off
    except ImportError:
        enc = [v.encode("utf-8") for v in values]
        off = np.zeros(len(enc) + 1, np.int64)
        np.cumsum([len(e) for e in enc], out=off[1:])
        return np.frombuffer(b"".join(enc), dtype=np.uint8), off


def column_key_bytes(col: pd.Series) -> tuple:
    """Canonical bytes of a key column for drop_duplicates-style equality (processor.py:140).

    -> (bytes u8, offsets i64, na_mask bool).  Equality classes match pandas': strings by value,
    all missing cells equal to each other (the caller overwrites their hash with NA_KEY), numeric
    columns by value with -0.0 == 0.0."""
    from . import pycells
    kind = col.dtype.kind
    if kind == "O" and pycells.available() and pycells.all_str(col.to_numpy()):
        na = np.zeros(len(col), bool)                       # every cell is a str (parallel header walk): no isna pass (60 ms per 1M cells)
    else:
        na = col.isna().to_numpy()
    if kind in "iub":
        raw = np.ascontiguousarray(col.to_numpy().astype(np.int64)).view(np.uint8)
        return raw, np.arange(len(col) + 1, dtype=np.int64) * 8, na
    if kind == "f":
        v = col.to_numpy().astype(np.float64) + 0.0          # -0.0 + 0.0 == +0.0
        v[na] = 0.0
        raw = np.ascontiguousarray(v).view(np.uint8)
        return raw, np.arange(len(col) + 1, dtype=np.int64) * 8, na
    # object column.  The usual case — every present cell is a str (URLs) — needs no per-cell Python: the str objects' UTF-8
    # buffers are gathered by worker threads (pycells), or pyarrow walks the objects in C; missing cells come out empty (the
    # caller gives them NA_KEY).
    if pycells.available() and col.dtype == object:
        try:
            flat = pycells.flat_utf8(col.to_numpy(), na)
        except UnicodeEncodeError:
            flat = None
        if flat is not None:
            return flat[0], flat[1], na
    try:
        import pyarrow as pa
        arr = pa.array(col.to_numpy(), type=pa.large_string(), from_pandas=True)
        bufs = arr.buffers()
        off = np.frombuffer(bufs[1], dtype=np.int64, count=len(arr) + 1)
        data = np.frombuffer(bufs[2], dtype=np.uint8, count=int(off[-1])) if bufs[2] is not None else np.zeros(0, np.uint8)
        return data, off, na
    except ImportError:
        pass
    except Exception:  # noqa: BLE001 - pa.ArrowInvalid / ArrowTypeError: some cell is not a str -> the tagged spelling below
        pass
    # Mixed column.  A str keeps its plain utf-8 bytes (the same key as on the fast path above — shards of one table must
    # agree); every other value is tagged with a 0xFF byte, which no utf-8 text contains, so "1.0" the string and 1.0 the
    # number stay apart.
    vals = col.tolist()
    plain = all(m or type(v) is str or isinstance(v, (bool, int, float)) for v, m in zip(vals, na.tolist()))
    ids = {}                 # any other hashable: one id per equality class, by Python's own hash / == (what pandas' object
    cells = []               # table uses, so 1 == 1.0 == Decimal(1) fall together); an unhashable cell raises TypeError like pandas
    for v, missing in zip(vals, na.tolist()):
        if missing:
            cells.append(b"")
        elif type(v) is str:
            cells.append(v.encode("utf-8"))
        elif plain:
            # 1 == 1.0 == True hash alike in pandas' object table; an int that no float equals keeps its own spelling
            try:
                as_float = float(v)
                exact = not isinstance(v, int) or int(as_float) == v
            except OverflowError:
                exact = False
            cells.append((b"\xffn" + repr(as_float).encode()) if exact else (b"\xffi" + repr(int(v)).encode()))
        else:
            cells.append(b"\xffo%d" % ids.setdefault(v, len(ids)))
    off = np.zeros(len(cells) + 1, np.int64)
    np.cumsum([len(c) for c in cells], out=off[1:])
    blob = b"".join(cells)
    return (np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(0, np.uint8)), off, na


def column_str_bytes(col: pd.Series, drop_na: bool = False) -> tuple:
    """``col.astype(str)`` (after ``dropna`` when asked) as flat bytes (processor.py:194, :198)."""
    from . import pycells
    every_cell_str = col.dtype == object and pycells.available() and pycells.all_str(col.to_numpy())
    if drop_na and not every_cell_str:
        col = col.dropna()
    if pycells.available() and col.dtype == object:      # an all-str column: astype(str) changes nothing but the missing cells (NaN -> "nan", None -> "None")
        try:
            na = np.zeros(len(col), bool) if every_cell_str else col.isna().to_numpy()
            flat = pycells.flat_utf8(col.to_numpy(), na, na_as_text=True)
        except UnicodeEncodeError:
            flat = None
        if flat is not None:
            return flat
    data, off = _strings_to_bytes(col.astype(str).tolist())
    return data, off
