"""Seeded synthetic annotation tables for parity tests and bench.py.

The generator is array-first: it draws the flat SoA layout the device stage consumes
(points, point offsets per box, box offsets per image row, labels, URL ids) with numpy, and
``to_frame`` renders the SAME arrays as the JSON-string DataFrame the step functions take
(column names of core/processor.py:244 / :283 of the reference), so the in-memory and the
CSV path see identical data.

Shape of a row (reference README / processor.py:262-296): ``{"width": W, "height": H,
"objects": [{"name": label, "polygon": {"ptList": [{"x":..,"y":..}, ...]}, ...}]}``.
"""
from __future__ import annotations

import json
from dataclasses import dataclass

import numpy as np

SEED = 20260227
ANN_COL = "结果字段-目标检测标签配置"
N_LABELS = 20


@dataclass
class SynthTable:
    n_rows: int
    xy: np.ndarray          # [P, 2] f64 polygon points
    pt_off: np.ndarray      # [B+1] i32 first point of each box
    box_off: np.ndarray     # [N+1] i32 first box of each row
    label: np.ndarray       # [B] i32 label id (name "c<id>")
    int_row: np.ndarray     # [N] bool row's coordinates are integers
    url_id: np.ndarray      # [N] i64 -> "http://img.example/<id>.jpg"
    width: int = 1920
    height: int = 1080

    @property
    def n_boxes(self) -> int:
        return len(self.pt_off) - 1

    @property
    def n_points(self) -> int:
        return len(self.xy)


def _ragged_arange(counts: np.ndarray) -> np.ndarray:
    """concatenate(arange(c) for c in counts) without a Python loop."""
    total = int(counts.sum())
    starts = np.cumsum(counts) - counts
    return np.arange(total, dtype=np.int64) - np.repeat(starts, counts)


def generate(n_rows: int, seed: int = SEED, boxes_per_row: int | None = None,
             max_boxes: int = 32, dup_prob: float = 0.05, tie_prob: float = 0.001) -> SynthTable:
    """Draw ``n_rows`` image rows.

    boxes/row ~ U{1..max_boxes} (or exactly ``boxes_per_row`` for the dense stress config);
    points/box ~ U{3..12}; centre ~ U(0,1920)xU(0,1080); point = centre + U(-50,50)^2; half
    of the rows integer-valued, the rest rounded to 2 decimals.  With probability
    ``dup_prob`` a row's LAST box is a copy of its first shrunk by U(0,3)% in height (so the
    HIGH class is populated on both sides of thr=0.98), and with ``tie_prob`` the last two
    boxes are the exact-tie pair (0,0,100,100)/(0,0,100,98) whose IoU is exactly 0.98.
    """
    rng = np.random.default_rng(seed)
    if boxes_per_row is None:
        nb = rng.integers(1, max_boxes + 1, size=n_rows, dtype=np.int64)
    else:
        nb = np.full(n_rows, boxes_per_row, dtype=np.int64)
    box_off = np.zeros(n_rows + 1, np.int64)
    np.cumsum(nb, out=box_off[1:])
    B = int(box_off[-1])
    row_of_box = np.repeat(np.arange(n_rows, dtype=np.int64), nb)

    int_row = rng.random(n_rows) < 0.5
    dup_row = (rng.random(n_rows) < dup_prob) & (nb >= 2)
    tie_row = (rng.random(n_rows) < tie_prob) & (nb >= 2) & ~dup_row

    npts = rng.integers(3, 13, size=B, dtype=np.int64)
    last_box = box_off[1:] - 1
    first_box = box_off[:-1]
    # dup rows: last box takes the point count of the first; tie rows: last two boxes have 4 points
    npts[last_box[dup_row]] = npts[first_box[dup_row]]
    npts[last_box[tie_row]] = 4
    npts[last_box[tie_row] - 1] = 4
    pt_off = np.zeros(B + 1, np.int64)
    np.cumsum(npts, out=pt_off[1:])
    P = int(pt_off[-1])
    if P >= 2 ** 31:
        raise ValueError("point count exceeds int32 offsets; generate in chunks")

    box_of_pt = np.repeat(np.arange(B, dtype=np.int64), npts)
    cx = rng.random(B) * 1920.0
    cy = rng.random(B) * 1080.0
    xy = np.empty((P, 2), np.float64)
    xy[:, 0] = cx[box_of_pt] + (rng.random(P) * 100.0 - 50.0)
    xy[:, 1] = cy[box_of_pt] + (rng.random(P) * 100.0 - 50.0)

    # near-duplicate boxes: copy the first box's points, shrink towards its top edge
    if dup_row.any():
        src = first_box[dup_row]
        dst = last_box[dup_row]
        cnt = npts[src]
        k = _ragged_arange(cnt)
        s_idx = np.repeat(pt_off[src], cnt) + k
        d_idx = np.repeat(pt_off[dst], cnt) + k
        shrink = np.repeat(rng.random(len(src)) * 0.03, cnt)
        ymin = np.minimum.reduceat(xy[s_idx, 1], np.cumsum(cnt) - cnt)
        ymin = np.repeat(ymin, cnt)
        xy[d_idx, 0] = xy[s_idx, 0]
        xy[d_idx, 1] = ymin + (xy[s_idx, 1] - ymin) * (1.0 - shrink)

    int_pt = int_row[row_of_box][box_of_pt]
    xy[int_pt] = np.rint(xy[int_pt])
    xy[~int_pt] = np.round(xy[~int_pt], 2)

    if tie_row.any():
        a = np.array([[0, 0], [100, 0], [100, 100], [0, 100]], np.float64)
        b = np.array([[0, 0], [100, 0], [100, 98], [0, 98]], np.float64)
        for lb in last_box[tie_row]:
            xy[pt_off[lb - 1]:pt_off[lb - 1] + 4] = a
            xy[pt_off[lb]:pt_off[lb] + 4] = b

    label = rng.integers(0, N_LABELS, size=B, dtype=np.int64).astype(np.int32)
    url_id = rng.integers(0, max(1, int(0.9 * n_rows)) + 1, size=n_rows, dtype=np.int64)
    return SynthTable(n_rows=n_rows, xy=xy, pt_off=pt_off.astype(np.int32),
                      box_off=box_off.astype(np.int32), label=label, int_row=int_row,
                      url_id=url_id)


def generate_device(n_rows: int, seed: int, device, boxes_per_row: int | None = None, max_boxes: int = 32,
                    dup_prob: float = 0.05, tie_prob: float = 0.001) -> dict:
    """The same table shape as ``generate`` drawn ON the device with torch (for the 10M-row bench tables, where the numpy
    generator would take minutes): identical distributions and planted rows (near-duplicate last box, exact-tie pair,
    integer-valued half), a different random stream.  -> dict of device tensors xy [P,2] f64, pt_off [B+1] i32,
    box_off [N+1] i32, label [B] i32, int_row [N] bool, url_id [N] i64."""
    import torch

    g = torch.Generator(device=device).manual_seed(int(seed))
    kw = {"generator": g, "device": device}
    if boxes_per_row is None:
        nb = torch.randint(1, max_boxes + 1, (n_rows,), dtype=torch.int64, **kw)
    else:
        nb = torch.full((n_rows,), boxes_per_row, dtype=torch.int64, device=device)
    box_off = torch.zeros(n_rows + 1, dtype=torch.int64, device=device)
    box_off[1:] = torch.cumsum(nb, 0)
    B = int(box_off[-1])
    int_row = torch.rand(n_rows, **kw) < 0.5
    dup_row = (torch.rand(n_rows, **kw) < dup_prob) & (nb >= 2)
    tie_row = (torch.rand(n_rows, **kw) < tie_prob) & (nb >= 2) & ~dup_row
    npts = torch.randint(3, 13, (B,), dtype=torch.int64, **kw)
    last_box, first_box = box_off[1:] - 1, box_off[:-1]
    npts[last_box[dup_row]] = npts[first_box[dup_row]]
    npts[last_box[tie_row]] = 4
    npts[last_box[tie_row] - 1] = 4
    pt_off = torch.zeros(B + 1, dtype=torch.int64, device=device)
    pt_off[1:] = torch.cumsum(npts, 0)
    P = int(pt_off[-1])
    if P >= 2 ** 31:
        raise ValueError("point count exceeds int32 offsets; generate in chunks")
    box_of_pt = torch.repeat_interleave(torch.arange(B, device=device), npts)
    centre = torch.rand((B, 2), dtype=torch.float64, **kw) * torch.tensor([1920.0, 1080.0], dtype=torch.float64, device=device)
    # one 1-D gather per column: `centre[box_of_pt]` (and index_select) on the [B, 2] tensor return zeros past ~59 M gathered rows on
    # torch 2.10 + ROCm 7.0 (tools/rng_probe.py), which put every later box of a big table at the origin
    xy = torch.rand((P, 2), dtype=torch.float64, **kw) * 100.0 - 50.0
    for col in (0, 1):
        xy[:, col] += centre[:, col].contiguous()[box_of_pt]
    if P:   # the draw is only worth something if the gather did what it says: spot-check it
        probe = torch.randint(0, P, (4096,), device=device)
        probe[-1] = P - 1
        off = xy[probe] - centre[box_of_pt[probe]]
        if not bool(((off >= -50.0) & (off <= 50.0)).all()):
            raise RuntimeError("synth.generate_device: gathered centres do not match (device indexing fault)")
    del centre
    if bool(dup_row.any()):
        src, dst = first_box[dup_row], last_box[dup_row]
        cnt = npts[src]
        k = torch.arange(int(cnt.sum()), device=device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
        s_idx = torch.repeat_interleave(pt_off[src], cnt) + k
        d_idx = torch.repeat_interleave(pt_off[dst], cnt) + k
        seg = torch.repeat_interleave(torch.arange(len(src), device=device), cnt)
        ymin = torch.full((len(src),), float("inf"), dtype=torch.float64, device=device).scatter_reduce(0, seg, xy[s_idx, 1], "amin")
        shrink = torch.rand(len(src), dtype=torch.float64, **kw) * 0.03
        xy[d_idx, 0] = xy[s_idx, 0]
        xy[d_idx, 1] = ymin[seg] + (xy[s_idx, 1] - ymin[seg]) * (1.0 - shrink[seg])
    int_pt = int_row[torch.repeat_interleave(torch.arange(n_rows, device=device), nb)][box_of_pt]
    del box_of_pt
    xy = torch.where(int_pt[:, None], torch.round(xy), torch.round(xy * 100.0) / 100.0)
    del int_pt
    if bool(tie_row.any()):
        lb = last_box[tie_row]
        a = torch.tensor([[0, 0], [100, 0], [100, 100], [0, 100]], dtype=torch.float64, device=device)
        b = torch.tensor([[0, 0], [100, 0], [100, 98], [0, 98]], dtype=torch.float64, device=device)
        four = torch.arange(4, device=device)
        xy[(pt_off[lb - 1][:, None] + four).reshape(-1)] = a.repeat(len(lb), 1)
        xy[(pt_off[lb][:, None] + four).reshape(-1)] = b.repeat(len(lb), 1)
    label = torch.randint(0, N_LABELS, (B,), dtype=torch.int64, **kw).to(torch.int32)
    url_id = torch.randint(0, max(1, int(0.9 * n_rows)) + 1, (n_rows,), dtype=torch.int64, **kw)
    return {"xy": xy, "pt_off": pt_off.to(torch.int32), "box_off": box_off.to(torch.int32), "label": label,
            "int_row": int_row, "url_id": url_id}


def table_from_device(d: dict) -> SynthTable:
    """host copy of a generate_device result (e.g. to render its JSON cells)"""
    c = {k: v.cpu().numpy() for k, v in d.items()}
    return SynthTable(n_rows=len(c["int_row"]), xy=c["xy"], pt_off=c["pt_off"], box_off=c["box_off"], label=c["label"],
                      int_row=c["int_row"], url_id=c["url_id"])


def urls(t: SynthTable) -> list:
    return [f"http://img.example/{k}.jpg" for k in t.url_id.tolist()]


def reference_urls(n_rows: int) -> list:
    """The reference set: every URL id divisible by 10 (about 10 % of main rows hit)."""
    return [f"http://img.example/{k}.jpg" for k in range(0, int(0.9 * n_rows) + 1, 10)]


def rules() -> dict:
    """label -> category of SURVEY §8d: c0..c9 -> catA, c10..c17 -> catB, c18/c19 undefined."""
    m = {f"c{i}": "catA" for i in range(10)}
    m.update({f"c{i}": "catB" for i in range(10, 18)})
    return m


def row_json(t: SynthTable, r: int) -> str:
    as_int = bool(t.int_row[r])
    objs = []
    for b in range(int(t.box_off[r]), int(t.box_off[r + 1])):
        pts = t.xy[int(t.pt_off[b]):int(t.pt_off[b + 1])].tolist()
        if as_int:
            pl = [{"x": int(x), "y": int(y)} for x, y in pts]
        else:
            pl = [{"x": x, "y": y} for x, y in pts]
        objs.append({"id": b - int(t.box_off[r]), "name": f"c{int(t.label[b])}",
                     "polygon": {"ptList": pl}, "type": "polygon"})
    return json.dumps({"width": t.width, "height": t.height, "objects": objs}, ensure_ascii=False)


def json_buffers(t: SynthTable, n_threads: int = 0):
    """(utf-8 bytes u8, offsets i64 [n+1]) of every row's JSON cell, written natively by all cores (dyd_synth_json: host code of
    libdyd_gfx950.so, no GPU involved) — byte for byte what row_json gives."""
    import ctypes as C

    from . import _native

    L = _native.load_library()
    xy = np.ascontiguousarray(t.xy, dtype=np.float64)
    pt_off = np.ascontiguousarray(t.pt_off, dtype=np.int32)
    box_off = np.ascontiguousarray(t.box_off, dtype=np.int32)
    label = np.ascontiguousarray(t.label, dtype=np.int32)
    int_row = np.ascontiguousarray(t.int_row, dtype=np.uint8)
    off = np.zeros(t.n_rows + 1, np.int64)
    text = C.c_void_p()
    _native.check(L.dyd_synth_json(xy.ctypes.data, pt_off.ctypes.data, box_off.ctypes.data, label.ctypes.data, int_row.ctypes.data,
                                   t.n_rows, t.width, t.height, n_threads, C.byref(text), off.ctypes.data), "dyd_synth_json")
    try:
        data = np.ctypeslib.as_array(C.cast(text, C.POINTER(C.c_uint8)), shape=(max(int(off[-1]), 1),))[:int(off[-1])].copy()
    finally:
        L.dyd_host_free(text)
    return data, off


def json_cells(t: SynthTable) -> np.ndarray:
    """object array of the rows' JSON cells (str): native writer + native str creation when the library is built, the
    per-row Python rendering otherwise"""
    try:
        from . import native_json

        data, off = json_buffers(t)
        return native_json.strings_from_buffers(data, off)
    except Exception:  # noqa: BLE001 - library not built: the portable rendering
        out = np.empty(t.n_rows, object)
        out[:] = [row_json(t, r) for r in range(t.n_rows)]
        return out


def to_frame(t: SynthTable):
    """The JSON-string DataFrame of the same table."""
    import pandas as pd

    return pd.DataFrame({"source": urls(t), ANN_COL: json_cells(t)})
