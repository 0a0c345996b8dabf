"""Test-only helpers.  ``OracleBackend`` lets the CPU suite drive the product's HOST logic
(flatten / emit / sharding) with the CPU checker standing in for the device stage; it lives
under tests/ because only tests may touch oracle/."""
import io
import os

import numpy as np
import pandas as pd

from oracle import lib as olib

_KEEP = {"first": 0, "last": 1, False: 2}


class OracleBackend:
    name = "oracle"

    def bbox_minmax(self, xy, pt_off):
        return olib.bbox_minmax(xy, pt_off)

    def iou_any_ge(self, box4, row_off, min_boxes, thr, want_max=False):
        return olib.iou_any_ge(box4, row_off, min_boxes, thr, want_max)

    def bbox_iou_fused(self, xy, pt_off, box_off, min_boxes, thr, want_box=False):
        box, arg, high = olib.bbox_iou_chain(xy, pt_off, box_off, min_boxes, thr)
        return (arg, high, box) if want_box else (arg, high)

    def hash128(self, data, off):
        return olib.hash128(data, off)

    def dedup(self, h, keep):
        return olib.dedup(h, _KEEP[keep])

    def isin(self, h, ref_h):
        return olib.isin(h, ref_h)

    def mt19937_permutation(self, seed, n):
        return olib.mt19937_permutation(seed, n)

    def split_ids(self, cat, perm, cat_off, n_train, n_val):
        return olib.split_ids(cat, perm, cat_off, n_train, n_val)

    def yolo_lines(self, box4, row_off, sel, width, height, class_id):
        return olib.yolo_lines(box4, row_off, sel, width, height, class_id)


def write_csv_text(path, text):
    with open(path, "w", encoding="utf-8-sig", newline="") as f:
        f.write(text)


def read_text(path):
    with open(path, "rb") as f:
        return f.read().decode("utf-8-sig")


def frame_records(f: pd.DataFrame):
    import json
    return json.loads(f.to_json(orient="split", force_ascii=False))


def random_polygons(rng, n_boxes, max_pts=12, special=True):
    """Ragged random point sets incl. ties, -0.0, NaN-first, inf and empty boxes."""
    npts = rng.integers(0 if special else 1, max_pts + 1, size=n_boxes)
    off = np.zeros(n_boxes + 1, np.int32)
    np.cumsum(npts, out=off[1:])
    P = int(off[-1])
    xy = np.round(rng.random((P, 2)) * 40 - 20, 0 if special else 3)   # coarse grid -> many ties
    if special and P:
        k = rng.integers(0, P, size=max(1, P // 50))
        xy[k, 0] = -0.0
        k = rng.integers(0, P, size=max(1, P // 80))
        xy[k, 1] = np.nan
        k = rng.integers(0, P, size=max(1, P // 90))
        xy[k, 0] = np.inf
        k = rng.integers(0, P, size=max(1, P // 90))
        xy[k, 1] = -np.inf
        first = off[:-1][npts > 0]
        sel = first[rng.random(len(first)) < 0.05]
        xy[sel, 0] = np.nan                                         # NaN at position 0 poisons
    return xy, off


def random_boxes(rng, n_rows, max_boxes=32, special=True, fixed=None):
    nb = np.full(n_rows, fixed) if fixed is not None else rng.integers(0, max_boxes + 1, size=n_rows)
    off = np.zeros(n_rows + 1, np.int32)
    np.cumsum(nb, out=off[1:])
    B = int(off[-1])
    c = rng.random((B, 2)) * 200
    wh = rng.random((B, 2)) * 60 + 1
    box = np.concatenate([c, c + wh], axis=1)
    box = np.round(box, 0) if special else box
    # near duplicates / exact ties inside rows
    for r in rng.integers(0, n_rows, size=max(1, n_rows // 4)):
        s, e = off[r], off[r + 1]
        if e - s >= 2:
            i, j = rng.integers(s, e, size=2)
            if i != j:
                box[j] = box[i]
                mode = rng.integers(0, 4)
                if mode == 1:
                    box[j, 3] -= (box[j, 3] - box[j, 1]) * 0.02          # right at the 0.98 boundary
                elif mode == 2:
                    box[j, 3] -= 1.0
                elif mode == 3:
                    box[j] = box[j][[2, 3, 0, 1]]                        # un-normalised corners
    if special and B:
        k = rng.integers(0, B, size=max(1, B // 100))
        box[k, rng.integers(0, 4, size=len(k))] = np.nan
        k = rng.integers(0, B, size=max(1, B // 150))
        box[k, 2] = np.inf
        k = rng.integers(0, B, size=max(1, B // 150))
        box[k] = box[k][:, [0, 1, 0, 1]]                                 # zero-area boxes
    return box, off
