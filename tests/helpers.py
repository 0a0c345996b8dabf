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

    def dedup_partner(self, h):
        """first row with the same 128-bit key, per row (numpy; what dyd_dedup_partner computes on the device)"""
        keys = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2).view([("a", np.uint64), ("b", np.uint64)]).reshape(-1)
        _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
        return first[inverse].astype(np.int64)

    def isin_partner(self, h, ref_h):
        h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 2)
        ref_h = np.ascontiguousarray(ref_h, dtype=np.uint64).reshape(-1, 2)
        table = {}
        for j, k in enumerate(map(tuple, ref_h.tolist())):
            table.setdefault(k, j)
        return np.asarray([table.get(k, -1) for k in map(tuple, h.tolist())], np.int64)

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


def split_expected_tablewise(df, label_to_category, json_columns=None, train_ratio=0.8, val_ratio=0.1, test_ratio=0.1, random_seed=42):
    """What oracle.steps.split_frames returns, assembled table-wise: the same per-row walk over the oracle's own primitives
    (parse_objects, split_object_labels: processor.py:712-792) collects INDICES instead of ``row.copy()`` per record — the
    copies are what make the port walk 150 rows a second — and the frames come from one ``take`` per table; the shuffle is
    pandas' own ``sample(frac=1, random_state=seed)`` (:800).  tests/test_split_cpu.py pins this against split_frames itself."""
    import copy
    import json

    from oracle import steps as osteps

    s = train_ratio + val_ratio + test_ratio
    train_ratio, val_ratio = train_ratio / s, val_ratio / s
    if json_columns is None:
        json_columns = [c for c in (osteps.NEW_COL, osteps.ANN_COL) if c in df.columns]
    present = [c for c in json_columns if c in df.columns]
    cols = {c: df[c].tolist() for c in present}
    sources = df["source"].tolist() if "source" in df.columns else [None] * len(df)
    rec = {"src": [], "label": [], "cat": [], "text": [], "combo": []}
    unc = {"src": [], "why": [], "label": []}
    counts = []
    for ri in range(len(df)):
        cell = None
        for c in present:
            v = cols[c][ri]
            if isinstance(v, str) and v:
                cell = v
                break
        doc, objs, err = osteps.parse_objects(cell)
        if err or not objs:
            why = err or "标注字段objects为空"
            unc["src"].append(ri); unc["why"].append(why); unc["label"].append(None)
            counts.append((sources[ri], "", 0, "否", why))
            continue
        seen = set()
        for o in objs:
            if isinstance(o, dict) and o.get("name"):
                seen.update(osteps.split_object_labels(o.get("name")))
        combo = "，".join(sorted(seen)) if seen else ""
        n_out, reasons = 0, set()
        for o in objs:
            if not isinstance(o, dict):
                continue
            labs = osteps.split_object_labels(o.get("name"))
            if not labs:
                unc["src"].append(ri); unc["why"].append("标注框缺少name字段"); unc["label"].append(None)
                continue
            for lab in labs:
                if lab not in label_to_category:
                    unc["src"].append(ri); unc["why"].append(f"标签{lab}未在规则中定义"); unc["label"].append(lab)
                    reasons.add(f"标签{lab}未在规则中定义")
                    continue
                one = copy.deepcopy(o)
                one["name"] = lab
                slim = {k: v for k, v in doc.items() if k != "objects"}
                slim["objects"] = [one]
                rec["src"].append(ri); rec["label"].append(lab); rec["cat"].append(label_to_category[lab])
                rec["text"].append(json.dumps(slim, ensure_ascii=False)); rec["combo"].append(combo)
                n_out += 1
        if not n_out:
            unc["src"].append(ri); unc["label"].append(None)
            unc["why"].append("；".join(sorted(reasons)) if reasons else "标签无法匹配规则")
        counts.append((sources[ri], combo, n_out, "否" if not n_out else ("部分可分类" if reasons else "是"), "；".join(sorted(reasons))))
    src = np.asarray(rec["src"], np.int64)
    cat = np.asarray(rec["cat"], object)
    out, cat_counts = {}, {}
    for name in pd.unique(cat) if len(cat) else []:
        m = np.flatnonzero(cat == name)
        f = df.iloc[src[m]].copy()
        text = pd.Series([rec["text"][k] for k in m.tolist()], index=f.index, dtype=object)
        for c in present:
            f[c] = text
        f["分类标签"] = [rec["label"][k] for k in m.tolist()]
        f["分类类别"] = name
        f["原始标签组合"] = [rec["combo"][k] for k in m.tolist()]
        f = f.sample(frac=1, random_state=random_seed).reset_index(drop=True)
        a, b = int(len(f) * train_ratio), int(len(f) * val_ratio)
        out[name] = (f.iloc[:a], f.iloc[a:a + b], f.iloc[a + b:])
        cat_counts[name] = len(f)
    if unc["src"]:
        u = df.iloc[np.asarray(unc["src"], np.int64)].copy()
        u["无法分类原因"] = unc["why"]
        if any(v is not None for v in unc["label"]):
            u["无法分类标签"] = [np.nan if v is None else v for v in unc["label"]]
    else:
        u = pd.DataFrame()
    c = pd.DataFrame(counts, columns=["source", "原始标签组合", "拆分条数", "是否可分类", "无法分类原因"]) if counts else pd.DataFrame()
    return {"categories": out, "unclassified": u, "split_counts": c, "category_counts": cat_counts}
