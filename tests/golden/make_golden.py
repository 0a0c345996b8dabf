#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where the reference is mounted read-only at
/root/reference (it never travels to the GPU box; the fixtures do).  The reference ships no
tests or vectors of its own, so these outputs are what pins oracle/ and the HIP path:

    replace_cases.json   a3 process_csv_replace_ptlist  edge-case matrix (value cases + raising cases)
    iou_cases.json       a4 filter_by_box_count_and_iou edge-case matrix
    dedup_cases.json     a1 deduplicate_csv_by_source   keep = first / last / False, NaN sources
    ref_filter_cases.json a2 remove_duplicates_between_csv NaN / "nan" / numeric columns
    perm_cases.json      a5 DataFrame.sample(frac=1, random_state=seed) orders + int(n*ratio) cuts
    split_case.json      a5 split_dataset_by_rules with Excel I/O captured in memory
    e2e_*.csv.gz         one seeded table through all five steps (inputs and every output)
    yolo_cases.json      f4 _extract_boxes_with_labels + the label files generate_yolo_datasets_from_excels writes
    merge_case.json      f3 merge_all_csv_in_folder: input files, merged bytes, progress-callback arguments, printed lines
    label_replace_case.json  replace_labels_by_mapping (the step between a4 and a5): output CSV, summary, diff / unmatched sheets, raising cells
    draw_case.json       download_and_draw_annotations (pipeline step "download") on images already on disk: the annotated PNGs
    summaries_case.json  summarize_unclassified (three sheets) and summarize_yolo_label_counts (stats + flat rows) on small inputs

Usage:  python tests/golden/make_golden.py [name ...]      (no names: everything)
"""
import gzip
import io
import json
import os
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import src.deal_yolo_data.core.processor as ref  # noqa: E402  (the reference itself)
from deal_yolo_daya_amd import synth  # noqa: E402

ANN = "结果字段-目标检测标签配置"
NEW = "新_" + ANN


def _dump(name, obj):
    with open(os.path.join(HERE, name), "w", encoding="utf-8") as f:
        json.dump(obj, f, ensure_ascii=False, indent=1)
    print("wrote", name)


def _read_text(p):
    with open(p, "rb") as f:
        return f.read().decode("utf-8-sig")


# ------------------------------------------------------------------------------- a3
REPLACE_VALUE_CASES = {
    "int_polygon": '{"width": 640, "height": 480, "objects": [{"name": "a", "polygon": {"ptList": [{"x": 5, "y": 9}, {"x": 1, "y": 12}, {"x": 7, "y": 3}]}}]}',
    "float_polygon": '{"objects": [{"polygon": {"ptList": [{"x": 5.5, "y": 9.25}, {"x": 1.0, "y": 12.0}, {"x": 7.75, "y": 3.5}]}}]}',
    "int_float_tie_first_wins": '{"objects": [{"polygon": {"ptList": [{"x": 1, "y": 2.0}, {"x": 1.0, "y": 2}, {"x": 1, "y": 2}]}}, {"polygon": {"ptList": [{"x": 1.0, "y": 2}, {"x": 1, "y": 2.0}]}}]}',
    "neg_zero_tie": '{"objects": [{"polygon": {"ptList": [{"x": 0, "y": -0.0}, {"x": -0.0, "y": 0}, {"x": 0.0, "y": 0.0}]}}, {"polygon": {"ptList": [{"x": -0.0, "y": 0.0}, {"x": 0, "y": -0.0}]}}]}',
    "nan_first_poisons": '{"objects": [{"polygon": {"ptList": [{"x": NaN, "y": 1}, {"x": 2, "y": NaN}, {"x": 3, "y": 0}]}}]}',
    "nan_later_ignored": '{"objects": [{"polygon": {"ptList": [{"x": 2, "y": 5}, {"x": NaN, "y": NaN}, {"x": 3, "y": 4}]}}]}',
    "infinities": '{"objects": [{"polygon": {"ptList": [{"x": Infinity, "y": -Infinity}, {"x": 1e308, "y": -1e308}, {"x": -Infinity, "y": Infinity}]}}]}',
    "empty_ptlist_gives_null": '{"objects": [{"name": "a", "polygon": {"ptList": []}}]}',
    "missing_polygon_added": '{"objects": [{"name": "a"}, {"name": "b", "polygon": {}}]}',
    "points_missing_keys_skipped": '{"objects": [{"polygon": {"ptList": [{"x": 1}, {"y": 2}, 5, "s", null, [1, 2], {"x": 9, "y": 8, "z": 7}]}}]}',
    "non_dict_objects_dropped": '{"objects": [1, "s", null, [1], {"polygon": {"ptList": [{"x": 1, "y": 2}, {"x": 3, "y": 4}]}}]}',
    "key_order_and_extras_kept": '{"b": [1, {"k": "v"}], "objects": [{"z": 1, "polygon": {"q": true, "ptList": [{"x": 1, "y": 2}], "r": null}, "a": "中文 \\" \\\\ \\u00e9 \\n"}], "a": 1.50}',
    "no_objects_key": '{"width": 10}',
    "objects_empty": '{"objects": [], "height": 5}',
    "single_point": '{"objects": [{"polygon": {"ptList": [{"x": 3, "y": 4}]}}]}',
    "big_ints_exact": '{"objects": [{"polygon": {"ptList": [{"x": 9007199254740993, "y": 1}, {"x": 9007199254740992, "y": 2}, {"x": 9007199254740994, "y": 0}]}}]}',
    "bool_coords": '{"objects": [{"polygon": {"ptList": [{"x": true, "y": 0}, {"x": 0, "y": false}, {"x": 1, "y": 1}]}}]}',
    "string_coords_compare": '{"objects": [{"polygon": {"ptList": [{"x": "b", "y": "10"}, {"x": "a", "y": "9"}]}}]}',
    "ptlist_is_dict": '{"objects": [{"polygon": {"ptList": {"x": 1, "y": 2}}}]}',
    "ptlist_is_string": '{"objects": [{"polygon": {"ptList": "abc"}}]}',
    "number_formats": '{"objects": [{"polygon": {"ptList": [{"x": 1E2, "y": 1.0e-5}, {"x": 100.0, "y": 0.00001}, {"x": 12345678901234567890, "y": 1e22}, {"x": 1.5e300, "y": 123456789.123456789}]}}], "width": 1e16, "height": -0}',
    "undecodable_json": '{"objects": [',
    "not_json_at_all": 'hello',
    "empty_string_cell": ' ',
    "dup_keys_last_wins": '{"objects": [{"polygon": {"ptList": [{"x": 1, "y": 2, "x": 5}]}, "polygon2": 1}], "width": 1, "width": 2}',
    "unicode_escapes": '{"objects": [{"name": "\\u4e2d\\u6587\\ud83d\\ude00/\\/", "polygon": {"ptList": [{"x": 1, "y": 1}]}}]}',
    "width_height_types": '{"width": "1920", "height": null, "objects": []}',
}
REPLACE_RAISING_CASES = {
    "coord_null": '{"objects": [{"polygon": {"ptList": [{"x": null, "y": 1}, {"x": 2, "y": 3}]}}]}',
    "polygon_null": '{"objects": [{"polygon": null}]}',
    "top_level_list": '[1, 2]',
    "objects_null": '{"objects": null}',
    "ptlist_null": '{"objects": [{"polygon": {"ptList": null}}]}',
    "ptlist_number": '{"objects": [{"polygon": {"ptList": 5}}]}',
    "mixed_str_int": '{"objects": [{"polygon": {"ptList": [{"x": "a", "y": 1}, {"x": 2, "y": 3}]}}]}',
    "polygon_is_list": '{"objects": [{"polygon": [1, 2]}]}',
    "top_level_number": '5',
    "coord_list_vs_int": '{"objects": [{"polygon": {"ptList": [{"x": [1], "y": 1}, {"x": 2, "y": 3}]}}]}',
}


def _run_replace(cells, sources=None):
    with tempfile.TemporaryDirectory() as d:
        df = pd.DataFrame({"source": sources or [f"u{i}" for i in range(len(cells))], ANN: cells,
                           "extra": list(range(len(cells)))})
        inp, out, exc = (os.path.join(d, n) for n in ("in.csv", "out.csv", "exc.csv"))
        df.to_csv(inp, index=False, encoding="utf-8-sig")
        res = ref.process_csv_replace_ptlist(inp, out, exc)
        return res, _read_text(inp), _read_text(out), _read_text(exc)


def make_replace():
    names = list(REPLACE_VALUE_CASES)
    cells = [REPLACE_VALUE_CASES[n] for n in names] + [None]     # a NaN annotation -> excluded
    res, inp, out, exc = _run_replace(cells, names + ["nan_annotation_row"])
    odf = pd.read_csv(io.StringIO(out))
    per = {}
    for i, n in enumerate(names):
        row = odf[odf["source"] == n].iloc[0]
        per[n] = {"in": REPLACE_VALUE_CASES[n],
                  "out": None if pd.isna(row[NEW]) else row[NEW],
                  "width": None if pd.isna(row["width"]) else row["width"],
                  "height": None if pd.isna(row["height"]) else row["height"]}
    raising = {}
    for n, cell in REPLACE_RAISING_CASES.items():
        try:
            _run_replace([REPLACE_VALUE_CASES["int_polygon"], cell])
            raising[n] = {"in": cell, "raises": None}
        except Exception as e:  # noqa: BLE001
            raising[n] = {"in": cell, "raises": type(e).__name__}
    _dump("replace_cases.json", {"result": {"filtered_rows": res["filtered_rows"], "excluded_rows": res["excluded_rows"]},
                                 "input_csv": inp, "output_csv": out, "excluded_csv": exc,
                                 "value_cases": per, "raising_cases": raising})


# ------------------------------------------------------------------------------- a4
def _bx(*boxes):
    return json.dumps({"objects": [{"polygon": {"ptList": [{"x": b[0], "y": b[1]}, {"x": b[2], "y": b[3]}]}} for b in boxes]})


IOU_CASES = {
    "identical_boxes": _bx((0, 0, 10, 10), (0, 0, 10, 10)),
    "exact_tie_098_int": _bx((0, 0, 100, 100), (0, 0, 100, 98)),
    "exact_tie_098_float": _bx((0.0, 0.0, 100.0, 100.0), (0.0, 0.0, 100.0, 98.0)),
    "just_below": _bx((0, 0, 100, 100), (0, 0, 100, 97)),
    "just_below_float": _bx((0, 0, 100, 100), (0, 0, 100, 97.99999999)),
    "disjoint": _bx((0, 0, 10, 10), (20, 20, 30, 30)),
    "touching_edges": _bx((0, 0, 10, 10), (10, 0, 20, 10)),
    "one_box_only": _bx((0, 0, 10, 10)),
    "no_boxes": '{"objects": []}',
    "third_pair_hits": _bx((0, 0, 10, 10), (50, 50, 60, 60), (50, 50, 60, 60.1)),
    "unnormalised_corners": _bx((10, 10, 0, 0), (0, 10, 10, 0)),
    "zero_area_identical": _bx((5, 5, 5, 5), (5, 5, 5, 5)),
    "zero_area_line": _bx((0, 0, 10, 0), (0, 0, 10, 0)),
    "null_box_middle_truncates": '{"objects": [' + ",".join([
        '{"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}',
        '{"polygon": {"ptList": [{"x": null, "y": null}, {"x": null, "y": null}]}}',
        '{"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}']) + ']}',
    "null_box_last": '{"objects": [' + ",".join([
        '{"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}',
        '{"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}',
        '{"polygon": {"ptList": [{"x": null, "y": null}, {"x": null, "y": null}]}}']) + ']}',
    "three_point_ptlist_skipped": '{"objects": [{"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}, {"x": 1, "y": 1}]}}, {"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}, {"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}]}',
    "guard_fail_skipped_not_truncated": '{"objects": [{"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}, 7, {"polygon": {"ptList": [{"x": 0}, {"x": 10, "y": 10}]}}, {"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}]}',
    "ptlist_null_truncates": '{"objects": [{"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}, {"polygon": {"ptList": null}}, {"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}]}',
    "nan_coords": '{"objects": [{"polygon": {"ptList": [{"x": NaN, "y": 0}, {"x": 10, "y": 10}]}}, {"polygon": {"ptList": [{"x": 0, "y": 0}, {"x": 10, "y": 10}]}}]}',
    "inf_coords": '{"objects": [{"polygon": {"ptList": [{"x": -Infinity, "y": 0}, {"x": Infinity, "y": 10}]}}, {"polygon": {"ptList": [{"x": -Infinity, "y": 0}, {"x": Infinity, "y": 10}]}}]}',
    "mixed_int_float": _bx((0, 0, 100, 100), (0.0, 0.5, 100.0, 99.5)),
    "float_rounding": _bx((0.1, 0.2, 100.3, 100.7), (0.1, 0.2, 100.3, 98.69)),
    "undecodable": '{"objects": [',
    "top_level_list": '[1]',
    "bool_coords": _bx((False, False, True, True), (0, 0, 1, 1)),
    "big_ints": _bx((0, 0, 3037000500, 3037000500), (0, 0, 3037000500, 3037000499)),
    "negative_coords": _bx((-100, -100, -1, -1), (-100, -100, -1, -2)),
}
IOU_RAISING = {
    "string_coords": _bx(("a", "b", "c", "d"), ("a", "b", "c", "d")),
}
IOU_PARAMS = [(2, 0.98), (3, 0.98), (2, 0.0), (2, -1.0), (1, 0.5), (2, 1.0), (0, 0.98), (2, 0.9800000000000001)]


def _run_iou(cells, min_boxes, thr):
    with tempfile.TemporaryDirectory() as d:
        df = pd.DataFrame({"source": [f"u{i}" for i in range(len(cells))], NEW: cells})
        inp, hi, lo = (os.path.join(d, n) for n in ("in.csv", "hi.csv", "lo.csv"))
        df.to_csv(inp, index=False, encoding="utf-8-sig")
        ref.filter_by_box_count_and_iou(inp, hi, lo, min_boxes, thr)
        hs = set(pd.read_csv(hi)["source"]) if os.path.getsize(hi) > 4 else set()
        return [1 if f"u{i}" in hs else 0 for i in range(len(cells))], _read_text(inp), _read_text(hi), _read_text(lo)


def make_iou():
    names = list(IOU_CASES)
    cells = [IOU_CASES[n] for n in names] + [None]
    out = {"names": names + ["nan_cell"], "cells": cells, "runs": []}
    for mb, thr in IOU_PARAMS:
        mask, inp, hi, lo = _run_iou(cells, mb, thr)
        run = {"min_boxes": mb, "thr": thr, "high": mask}
        if (mb, thr) == (2, 0.98):
            run.update({"input_csv": inp, "high_csv": hi, "other_csv": lo})
        out["runs"].append(run)
    out["raising"] = {}
    for n, cell in IOU_RAISING.items():
        try:
            _run_iou([cell], 2, 0.98)
            out["raising"][n] = {"in": cell, "raises": None}
        except Exception as e:  # noqa: BLE001
            out["raising"][n] = {"in": cell, "raises": type(e).__name__}
    _dump("iou_cases.json", out)


# ------------------------------------------------------------------------------- a3 -> a4 in sequence
def _poly(*pts):
    return {"polygon": {"ptList": [{"x": x, "y": y} for x, y in pts]}}


_A = [(0, 0), (100, 0), (100, 100), (0, 100)]
_A98 = [(0, 0), (100, 0), (100, 98), (0, 98)]
_AF = [(0.5, 0.25), (100.5, 0.25), (100.5, 100.25), (0.5, 100.25)]
_FAR = [(500, 500), (600, 500), (600, 650)]
_BIG = [(2 ** 26, 0), (2 ** 26 + 100, 0), (2 ** 26 + 100, 100)]
CHAIN_CASES = {
    "A_null_A": {"objects": [_poly(*_A), _poly(), _poly(*_A)]},
    "A_A_null": {"objects": [_poly(*_A), _poly(*_A), _poly()]},
    "null_A_A": {"objects": [_poly(), _poly(*_A), _poly(*_A)]},
    "A_A98_exact_tie": {"objects": [_poly(*_A), _poly(*_A98)]},
    "A_far_A98_far": {"width": 1920, "height": 1080, "objects": [_poly(*_A), _poly(*_FAR), _poly(*_A98), _poly(*_FAR)]},
    "float_pair": {"width": 1920.5, "objects": [_poly(*_AF), _poly(*_AF)]},
    "int_float_mix": {"objects": [_poly(*_A), _poly(*[(float(x), float(y)) for x, y in _A])]},
    "single": {"objects": [_poly(*_A)]},
    "none": {"objects": []},
    "no_polygon_member_then_pair": {"objects": [{"name": "x"}, _poly(*_A), _poly(*_A)]},
    "pair_then_no_polygon_member": {"objects": [_poly(*_A), _poly(*_A), {"name": "x"}]},
    "points_without_keys_only": {"objects": [_poly(*_A), {"polygon": {"ptList": [{"x": 1}, 5, None]}}, _poly(*_A)]},
    "non_dict_objects": {"objects": [3, _poly(*_A), "s", _poly(*_A)]},
    "big_ints_pair": {"objects": [_poly(*_BIG), _poly(*_BIG)]},
    "big_ints_disjoint": {"objects": [_poly(*_BIG), _poly(*_FAR)]},
    "nan_first_point": {"objects": [_poly((float("nan"), 0), (100, 100)), _poly(*_A), _poly(*_A)]},
    "zero_area_identical": {"objects": [_poly((5, 5), (5, 5)), _poly((5, 5), (5, 5))]},
    "three_of_which_last_two": {"objects": [_poly(*_FAR), _poly(*_A), _poly(*_A98)]},
}
CHAIN_RAW = {"undecodable": '{"objects": [', "blank": " "}
CHAIN_PARAMS = [(2, 0.98), (3, 0.98), (2, 0.0), (1, 0.5), (2, 0.9800000000000001)]


def make_chain():
    """polygon cells through process_csv_replace_ptlist and then filter_by_box_count_and_iou, as the processing page runs
    them (ui/pages/processing.py:580-598): every file of both steps"""
    names = list(CHAIN_CASES) + list(CHAIN_RAW) + ["nan_annotation"]
    cells = [json.dumps(CHAIN_CASES[n], ensure_ascii=False) for n in CHAIN_CASES] + list(CHAIN_RAW.values()) + [None]
    sources = [f"http://img/{n}.jpg" for n in names]
    sources[3] = "007"            # read back as the number 7 by the IoU step
    out = {"names": names, "runs": []}
    with tempfile.TemporaryDirectory() as d:
        inp, proc, exc = (os.path.join(d, n) for n in ("in.csv", "processed.csv", "excluded.csv"))
        pd.DataFrame({"source": sources, ANN: cells, "extra": list(range(len(cells)))}).to_csv(inp, index=False, encoding="utf-8-sig")
        res = ref.process_csv_replace_ptlist(inp, proc, exc)
        out.update({"input_csv": _read_text(inp), "processed_csv": _read_text(proc), "excluded_csv": _read_text(exc),
                    "result": {"filtered_rows": res["filtered_rows"], "excluded_rows": res["excluded_rows"]}})
        for mb, thr in CHAIN_PARAMS:
            hi, lo = os.path.join(d, "hi.csv"), os.path.join(d, "lo.csv")
            ref.filter_by_box_count_and_iou(proc, hi, lo, mb, thr)
            out["runs"].append({"min_boxes": mb, "thr": thr, "high_csv": _read_text(hi), "other_csv": _read_text(lo)})
    _dump("chain_cases.json", out)


# ------------------------------------------------------------------------------- a1 / a2
def make_dedup():
    cases = {
        "strings_with_nan": "source,v\nu1,1\nu2,2\nu1,3\n,4\nu3,5\n,6\nu2,7\nU1,8\n u1,9\n",
        "all_unique": "source,v\na,1\nb,2\nc,3\n",
        "all_same": "source,v\na,1\na,2\na,3\n",
        "numeric_sources": "source,v\n1,1\n2,2\n1,3\n3,4\n",
        "float_sources_nan": "source,v\n1.5,1\n,2\n1.5,3\n,4\n0.0,5\n-0.0,6\n",
        "unicode": "source,v\n中文,1\n中文,2\nüber,3\nuber,4\n",
        "header_only": "source,v\n",
    }
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, text in cases.items():
            inp = os.path.join(d, name + ".csv")
            with open(inp, "w", encoding="utf-8-sig") as f:
                f.write(text)
            out[name] = {"input_csv": text, "keep": {}}
            for keep in ("first", "last", False):
                o = os.path.join(d, "o.csv")
                res = ref.deduplicate_csv_by_source(inp, o, keep=keep, verbose=False)
                out[name]["keep"][str(keep)] = {"output_csv": _read_text(o), "rows": len(res)}
    _dump("dedup_cases.json", out)


def make_ref_filter():
    cases = {
        "strings": ("source,v\nu1,1\nu2,2\nu3,3\nu4,4\n", "source\nu2\nu4\nu9\n"),
        "main_nan_vs_ref_nan_dropped": ("source,v\nu1,1\n,2\nu3,3\n", "source\n\nu3\n"),
        "main_nan_vs_ref_literal_nan": ("source,v\nu1,1\n,2\nnan,3\n", "source,k\nnan,1\nx,2\n"),
        "int_vs_int": ("source,v\n1,1\n2,2\n3,3\n", "source\n2\n5\n"),
        "int_vs_float": ("source,v\n1,1\n2,2\n3,3\n", "source\n2.0\n\n"),
        "float_vs_float": ("source,v\n1.0,1\n2.5,2\n,3\n", "source\n2.5\n1\n"),
        "other_column": ("id,source\na,1\nb,2\nc,3\n", "id,z\nb,0\nq,0\n"),
        "empty_ref": ("source,v\nu1,1\n", "source\n"),
    }
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, (m, r) in cases.items():
            mp, rp, op = (os.path.join(d, n) for n in ("m.csv", "r.csv", "o.csv"))
            for p, t in ((mp, m), (rp, r)):
                with open(p, "w", encoding="utf-8-sig") as f:
                    f.write(t)
            col = "id" if name == "other_column" else "source"
            res = ref.remove_duplicates_between_csv(mp, rp, op, compare_col=col, verbose=False)
            out[name] = {"main_csv": m, "ref_csv": r, "compare_col": col, "output_csv": _read_text(op), "rows": len(res)}
    _dump("ref_filter_cases.json", out)


# ------------------------------------------------------------------------------- a5
def make_perm():
    out = {"perms": [], "cuts": []}
    for seed in (0, 42, 9999, 4294967295):
        for n in (1, 2, 7, 100, 5000):
            order = pd.DataFrame({"i": np.arange(n)}).sample(frac=1, random_state=seed)["i"].tolist()
            out["perms"].append({"seed": seed, "n": n, "order": order})
    for ratios in ((0.8, 0.1, 0.1), (8, 1, 1), (0.7, 0.2, 0.1), (1, 1, 1), (0.5, 0.3, 0.3)):
        s = ratios[0] + ratios[1] + ratios[2]
        tr, va = ratios[0] / s, ratios[1] / s
        for n in (0, 1, 2, 3, 7, 9, 10, 11, 99, 100, 101, 1000, 8918, 12345, 10 ** 7 + 3):
            out["cuts"].append({"ratios": list(ratios), "n": n, "n_train": int(n * tr), "n_val": int(n * va)})
    _dump("perm_cases.json", out)


class _Capture:
    """Stand-in for the Excel layer (openpyxl is not installed): captures frames in memory."""

    def __init__(self):
        self.sheets = {}
        self.current = None

    def writer(self, path, *a, **k):
        cap = self

        class W:
            def __enter__(self_w):
                cap.current = str(os.path.basename(path))
                return self_w

            def __exit__(self_w, *e):
                cap.current = None
                return False
        return W()


def _frame_records(f):
    return json.loads(f.to_json(orient="split", force_ascii=False))


def run_reference_split(df, rules_df, seed=42, ratios=(0.8, 0.1, 0.1), rule_mode="wide", label_col=None, category_col=None):
    cap = _Capture()
    orig = (ref.pd.read_excel, ref.pd.ExcelWriter, pd.DataFrame.to_excel)

    def to_excel(self, target, sheet_name="Sheet1", index=True, **k):
        key = cap.current if cap.current else str(os.path.basename(str(target)))
        cap.sheets.setdefault(key, {})[sheet_name] = self.copy()

    ref.pd.read_excel = lambda *a, **k: rules_df
    ref.pd.ExcelWriter = cap.writer
    pd.DataFrame.to_excel = to_excel
    try:
        with tempfile.TemporaryDirectory() as d:
            inp = os.path.join(d, "in.csv")
            rules = os.path.join(d, "rules.xlsx")
            open(rules, "wb").close()
            df.to_csv(inp, index=False, encoding="utf-8-sig")
            res = ref.split_dataset_by_rules(inp, rules, os.path.join(d, "out"), rule_mode, None, label_col, category_col, None,
                                             ratios[0], ratios[1], ratios[2], seed)
            summary = res["summary"]
            files = [os.path.basename(str(p)) for p in res["category_files"]]
    finally:
        ref.pd.read_excel, ref.pd.ExcelWriter, pd.DataFrame.to_excel = orig
    return cap.sheets, summary, files


def make_split():
    t = synth.generate(60, seed=7, max_boxes=6)
    df = synth.to_frame(t)
    # a few hand-made rows: multi-label names, undefined labels, missing names, broken cells
    extra = pd.DataFrame({"source": ["m1", "m2", "m3", "m4", "m5", "m6"], ANN: [
        '{"objects": [{"name": "c1,c12；c19", "polygon": {"ptList": [{"x": 1, "y": 2}, {"x": 3, "y": 4}]}}]}',
        '{"objects": [{"name": "", "polygon": {}}, {"polygon": {}}, {"name": "c18"}]}',
        '{"objects": []}',
        '{"objects": [',
        '{"objects": {"a": 1}}',
        '{"width": 5, "objects": [7, {"name": " c3 | c3 "}]}',
    ]})
    df = pd.concat([df, extra], ignore_index=True)
    df.loc[len(df)] = ["m7", np.nan]
    rules_df = pd.DataFrame({"catA": [",".join(f"c{i}" for i in range(5)), "c5;c6", "c7|c8，c9"],
                             "catB": ["c10,c11,c12,c13", "c14；c15", "c16,c17"]})
    sheets, summary, files = run_reference_split(df, rules_df)
    out = {"input": _frame_records(df), "rules": _frame_records(rules_df), "seed": 42, "ratios": [0.8, 0.1, 0.1],
           "summary": summary, "category_files": files,
           "sheets": {k: {s: _frame_records(f) for s, f in v.items()} for k, v in sheets.items()}}
    # the other rule sheet layout (:697-703): one (label, category) pair per row, with blanks / NaN / "nan" to be skipped,
    # plus non-default ratios (normalised by their sum, :673-676) and another seed
    two = pd.DataFrame({"标签": [" c1 ", "c2", None, "nan", "c12", "c3", "c19", ""], "类别": ["catA", " catA ", "catB", "catB", "catB", None, "NaN", "catC"],
                        "备注": list("abcdefgh")})
    sheets2, summary2, files2 = run_reference_split(df, two, seed=7, ratios=(6, 3, 1), rule_mode="two_column", label_col="标签", category_col="类别")
    out["two_column"] = {"rules": _frame_records(two), "seed": 7, "ratios": [6, 3, 1], "label_col": "标签", "category_col": "类别",
                         "summary": summary2, "category_files": files2,
                         "sheets": {k: {s: _frame_records(f) for s, f in v.items()} for k, v in sheets2.items()}}
    _dump("split_case.json", out)


# ------------------------------------------------------------------------------- e2e
def make_e2e(n_rows=240):
    t = synth.generate(n_rows, seed=synth.SEED, max_boxes=12, dup_prob=0.15, tie_prob=0.03)
    df = synth.to_frame(t)
    df.loc[5, ANN] = np.nan                      # excluded by the replace step
    df.loc[9, ANN] = '{"objects": ['             # undecodable -> new column empty, row kept
    ref_df = pd.DataFrame({"source": synth.reference_urls(n_rows)})
    with tempfile.TemporaryDirectory() as d:
        P = lambda n: os.path.join(d, n)  # noqa: E731
        df.to_csv(P("merged.csv"), index=False, encoding="utf-8-sig")
        ref_df.to_csv(P("ref.csv"), index=False, encoding="utf-8-sig")
        import contextlib
        prints = {}

        def logged(name, fn, *a, **k):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                out = fn(*a, **k)
            prints[name] = buf.getvalue().replace(d, "<TMP>")
            return out

        logged("dedup", ref.deduplicate_csv_by_source, P("merged.csv"), P("dedup.csv"))
        logged("ref_filter", ref.remove_duplicates_between_csv, P("dedup.csv"), P("ref.csv"), P("filtered.csv"))
        logged("replace", ref.process_csv_replace_ptlist, P("filtered.csv"), P("processed.csv"), P("excluded.csv"))
        logged("iou", ref.filter_by_box_count_and_iou, P("processed.csv"), P("high.csv"), P("other.csv"), 2, 0.98)
        logged("replace_missing_file", ref.process_csv_replace_ptlist, P("nope.csv"), P("x.csv"), P("y.csv"))
        logged("iou_missing_column", ref.filter_by_box_count_and_iou, P("ref.csv"), P("h2.csv"), P("o2.csv"))
        logged("replace_missing_column", ref.process_csv_replace_ptlist, P("ref.csv"), P("x.csv"), P("y.csv"))
        errors = {}

        def raised(name, fn, *a, **k):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    fn(*a, **k)
                errors[name] = None
            except Exception as e:  # noqa: BLE001
                errors[name] = [type(e).__name__, str(e).replace(d, "<TMP>")]

        open(P("a.txt"), "w").write("source\n1\n")
        open(P("nosource.csv"), "w").write("a,b\n1,2\n")
        raised("dedup_missing_file", ref.deduplicate_csv_by_source, P("missing.csv"))
        raised("dedup_not_csv", ref.deduplicate_csv_by_source, P("a.txt"))
        raised("dedup_no_source_column", ref.deduplicate_csv_by_source, P("nosource.csv"), None)
        raised("ref_missing_ref_file", ref.remove_duplicates_between_csv, P("nosource.csv"), P("missing.csv"))
        raised("ref_missing_main_file", ref.remove_duplicates_between_csv, P("missing.csv"), P("nosource.csv"))
        raised("ref_not_csv", ref.remove_duplicates_between_csv, P("a.txt"), P("nosource.csv"))
        raised("ref_no_column", ref.remove_duplicates_between_csv, P("nosource.csv"), P("nosource.csv"), P("o.csv"))
        raised("ref_no_column_in_ref", ref.remove_duplicates_between_csv, P("merged.csv"), P("nosource.csv"), P("o.csv"))
        raised("split_missing_input", ref.split_dataset_by_rules, P("missing.csv"), P("nosource.csv"), P("out"))
        raised("split_missing_rules", ref.split_dataset_by_rules, P("merged.csv"), P("missing.xlsx"), P("out"))
        prints["errors"] = errors
        _dump("e2e_prints.json", prints)
        for n in ("merged", "ref", "dedup", "filtered", "processed", "excluded", "high", "other"):
            with open(P(n + ".csv"), "rb") as f, gzip.GzipFile(os.path.join(HERE, f"e2e_{n}.csv.gz"), "wb", mtime=0) as g:
                g.write(f.read())
        other = pd.read_csv(P("other.csv"), encoding="utf-8-sig")
    rules_df = pd.DataFrame({"catA": [f"c{i}" for i in range(10)], "catB": [f"c{i}" for i in range(10, 18)] + [None, None]})
    sheets, summary, files = run_reference_split(other, rules_df)
    slim = {}
    for k, v in sheets.items():
        slim[k] = {}
        for s, f in v.items():
            keep = [c for c in ("source", "分类标签", "分类类别", "原始标签组合", "无法分类原因", "无法分类标签", "拆分条数", "是否可分类", NEW) if c in f.columns]
            slim[k][s] = _frame_records(f[keep])
    _dump("e2e_split.json", {"summary": summary, "category_files": files, "rules": _frame_records(rules_df), "sheets": slim})
    print("wrote e2e_*.csv.gz")



# ------------------------------------------------------------------------------- f4  YOLO label lines
def _obj(name, *pts, raw=None):
    if raw is not None:
        return raw
    pl = ", ".join("{" + ", ".join(f'"{k}": {v}' for k, v in p.items()) + "}" if isinstance(p, dict) else str(p) for p in pts)
    head = f'"name": {json.dumps(name, ensure_ascii=False)}, ' if name is not None else ""
    return "{" + head + '"polygon": {"ptList": [' + pl + "]}}"


def _cell(*objs):
    return '{"objects": [' + ", ".join(objs) + "]}"


def P2(x1, y1, x2, y2):
    return ({"x": x1, "y": y1}, {"x": x2, "y": y2})


# (json cell, label value, width, height)
YOLO_CASES = {
    "int_box": (_cell(_obj("a", *P2(10, 20, 110, 220))), "a", 1920, 1080),
    "float_box": (_cell(_obj("a", *P2(10.25, 20.5, 110.75, 220.13))), "a", 1920, 1080),
    "float_size": (_cell(_obj("b", *P2(3, 4, 1000, 700))), "b", 1920.0, 1080.5),
    "polygon_many_points": (_cell(_obj("a", {"x": 5, "y": 9}, {"x": 1, "y": 12}, {"x": 7, "y": 3}, {"x": 6.5, "y": 30})), "a", 64, 48),
    "three_objects_two_match": (_cell(_obj("a", *P2(0, 0, 10, 10)), _obj("b", *P2(1, 1, 5, 5)), _obj("a", *P2(20, 30, 25.5, 31))), "a", 100, 50),
    "zero_width_only": (_cell(_obj("a", *P2(5, 0, 5, 10))), "a", 100, 100),
    "zero_height_mixed": (_cell(_obj("a", *P2(5, 7, 9, 7)), _obj("a", *P2(1, 2, 3, 4))), "a", 100, 100),
    "single_point_polygon": (_cell(_obj("a", {"x": 3, "y": 4})), "a", 100, 100),
    "nan_first_x": (_cell(_obj("a", {"x": "NaN", "y": 1}, {"x": 2, "y": 5})), "a", 100, 100),
    "nan_later_ignored": (_cell(_obj("a", {"x": 2, "y": 1}, {"x": "NaN", "y": "NaN"}, {"x": 4, "y": 5})), "a", 100, 100),
    "infinite_corner": (_cell(_obj("a", {"x": 1, "y": 1}, {"x": "Infinity", "y": 5})), "a", 100, 100),
    "both_infinite": (_cell(_obj("a", {"x": "-Infinity", "y": 1}, {"x": "Infinity", "y": 5})), "a", 100, 100),
    "negative_coords": (_cell(_obj("a", *P2(-50, -20.5, -10, 30))), "a", 640, 480),
    "negative_rounds_to_minus_zero": (_cell(_obj("a", *P2(-1e-9, -3e-10, 1e-10, 1e-10))), "a", 1920, 1080),
    "neg_zero_corners": (_cell(_obj("a", *P2(-0.0, -0.0, 0.0, 0.0))), "a", 10, 10),
    "tie_half_even_down": (_cell(_obj("a", *P2(0, 0, 2, 6))), "a", 128, 128),
    "tie_half_even_up": (_cell(_obj("a", *P2(0, 0, 6, 10))), "a", 128, 256),
    "just_above_tie": (_cell(_obj("a", *P2(0, 0, 2.0000000000000004, 6.000000000000001))), "a", 128, 128),
    "rounds_up_to_ten": (_cell(_obj("a", *P2(0, 0, 19.9999999, 39.99999999))), "a", 1, 2),
    "carry_into_integer_digits": (_cell(_obj("a", *P2(0, 0, 1.9999996, 1999.9999996))), "a", 1, 1),
    "large_normalised": (_cell(_obj("a", *P2(0, 0, 123456789012.25, 8796093022207.5))), "a", 1, 1),
    "at_2_pow_43": (_cell(_obj("a", *P2(0, 0, 17592186044416, 17592186044418))), "a", 1, 2),
    "huge_values": (_cell(_obj("a", *P2(0, 0, 1e300, 2.5e15))), "a", 1e-5, 3),
    "tiny_size_divisor": (_cell(_obj("a", *P2(1, 2, 3, 4))), "a", 1e-300, 5e-324),
    "denormal_results": (_cell(_obj("a", *P2(0, 0, 1e-310, 3e-320))), "a", 7, 1e5),
    "nan_width": (_cell(_obj("a", *P2(1, 2, 3, 4))), "a", float("nan"), 100),
    "inf_height": (_cell(_obj("a", *P2(1, 2, 3, 4))), "a", 100, float("inf")),
    "negative_width": (_cell(_obj("a", *P2(1, 2, 3, 4))), "a", -100, 100),
    "zero_width_image": (_cell(_obj("a", *P2(1, 2, 3, 4))), "a", 0, 100),
    "zero_float_height_image": (_cell(_obj("a", *P2(1, 2, 3, 4))), "a", 100, 0.0),
    "label_mismatch": (_cell(_obj("b", *P2(1, 2, 3, 4))), "a", 100, 100),
    "multi_label_name_never_equal": (_cell(_obj("a,b", *P2(1, 2, 3, 4))), "a", 100, 100),
    "escaped_name_equal": (_cell(_obj(None, raw='{"name": "\\u4e2d\\u6587", "polygon": {"ptList": [{"x": 1, "y": 2}, {"x": 3, "y": 4}]}}')), "中文", 100, 100),
    "x_and_y_lists_independent": (_cell(_obj("a", {"x": 1}, {"y": 2}, {"x": 5, "y": 7}, {"y": 9, "k": 0})), "a", 10, 10),
    "only_x_keys": (_cell(_obj("a", {"x": 1}, {"x": 2})), "a", 10, 10),
    "exception_keeps_earlier_boxes": (_cell(_obj("a", *P2(1, 2, 3, 4)), _obj("a", {"x": '"s"', "y": 1}, {"x": 2, "y": 2}), _obj("a", *P2(5, 6, 7, 8))), "a", 10, 10),
    "null_x_raises": (_cell(_obj("a", *P2(1, 2, 3, 4)), _obj("a", {"x": "null", "y": 1}, {"x": 2, "y": 2})), "a", 10, 10),
    "non_dict_points_and_objects": (_cell("7", '"s"', _obj("a", 5, '"t"', "null", {"x": 1, "y": 2}, {"x": 3, "y": 4})), "a", 10, 10),
    "ptlist_not_a_list": (_cell(_obj("a", *P2(1, 2, 3, 4)), '{"name": "a", "polygon": {"ptList": 5}}', _obj("a", *P2(5, 6, 7, 8))), "a", 10, 10),
    "polygon_not_a_dict": (_cell(_obj("a", *P2(1, 2, 3, 4)), '{"name": "a", "polygon": [1]}', _obj("a", *P2(5, 6, 7, 8))), "a", 10, 10),
    "bool_coordinates": (_cell(_obj("a", {"x": "true", "y": "false"}, {"x": 5, "y": 3})), "a", 10, 10),
    "big_int_coordinates": (_cell(_obj("a", *P2(9007199254740993, 1, 9007199254740997, 3))), "a", 4, 4),
    "int_beyond_2_pow_52": (_cell(_obj("a", *P2(4503599627370497, 1, 4503599627370499, 3))), "a", 3, 4),
    "numeric_name": (_cell(_obj(7, *P2(1, 2, 3, 4))), "7", 10, 10),
    "name_missing_or_empty": (_cell(_obj(None, *P2(1, 2, 3, 4)), _obj("", *P2(1, 2, 3, 4))), "a", 10, 10),
    "objects_not_a_list": ('{"objects": {"name": "a"}}', "a", 10, 10),
    "top_level_list": ('[1, 2]', "a", 10, 10),
    "undecodable": ('{"objects": [', "a", 10, 10),
    "duplicate_keys_last_wins": (_cell('{"name": "b", "name": "a", "polygon": {"ptList": [{"x": 1, "y": 2, "x": 4}, {"x": 3, "y": 4}]}}'), "a", 10, 10),
    "two_digit_class": (_cell(_obj("zz", *P2(1, 2, 3, 4))), "zz", 10, 10),
}


def run_reference_yolo(frames, seed=42, **kwargs):
    """generate_yolo_datasets_from_excels on in-memory sheets ({split: frame}), Excel layer stubbed, images
    local files -> everything the call leaves behind"""
    orig = (ref.pd.ExcelFile, ref.pd.read_excel, pd.DataFrame.to_excel)
    skipped_frames = []

    class XF:
        def __init__(self, path):
            self.sheet_names = list(frames)

    def to_excel(self, target, *a, **k):
        skipped_frames.append(self.copy())

    ref.pd.ExcelFile = XF
    ref.pd.read_excel = lambda path, sheet_name=None, **k: frames[sheet_name].copy()
    pd.DataFrame.to_excel = to_excel
    try:
        with tempfile.TemporaryDirectory() as d:
            book = os.path.join(d, "catA.xlsx")
            open(book, "wb").close()
            res = ref.generate_yolo_datasets_from_excels([book], os.path.join(d, "out"), download_images=False, random_seed=seed, **kwargs)
            ds = str(res["datasets"][0])
            labels, images = {}, {}
            for split in ("train", "val", "test"):
                labels[split], images[split] = {}, sorted(os.listdir(os.path.join(ds, "images", split)))
                for fn in sorted(os.listdir(os.path.join(ds, "labels", split))):
                    with open(os.path.join(ds, "labels", split, fn), "rb") as f:
                        labels[split][fn] = f.read().decode("utf-8")
            yaml_text = open(os.path.join(ds, "data.yaml"), encoding="utf-8").read().replace(ds, "<DATASET>")
            out = {"labels": labels, "images": images, "data_yaml": yaml_text, "stats": res["stats"], "total": res["total"],
                   "processed": res["processed"], "downloaded": res["downloaded"], "dataset_name_map": res["dataset_name_map"],
                   "skipped": json.loads(skipped_frames[-1].to_json(orient="records", force_ascii=False))}
    finally:
        ref.pd.ExcelFile, ref.pd.read_excel, pd.DataFrame.to_excel = orig
    return out


def make_yolo():
    import yaml
    import src.deal_yolo_data.core.utils as ref_utils
    with tempfile.TemporaryDirectory() as imgdir:
        rows = []
        for k, (name, (cell, label, w, h)) in enumerate(YOLO_CASES.items()):
            img = os.path.join(imgdir, f"case{k:03d}.jpg")
            open(img, "wb").write(b"x")
            rows.append({"source": img, "分类标签": label, NEW: cell, "width": w, "height": h})
        for k in range(12):                              # filler labels so that class ids reach two digits
            rows.append({"source": os.path.join(imgdir, "none.jpg"), "分类标签": f"k{k:02d}", NEW: _cell(), "width": 1, "height": 1})
        df = pd.DataFrame(rows)
        run = run_reference_yolo({"train": df})
        # a second, three-sheet run: fallback column, missing source / label, image-less rows, resume files
        t = synth.generate(40, seed=11, max_boxes=3)
        base = synth.to_frame(t)
        recs = []
        for i in range(len(base)):
            doc = json.loads(base.loc[i, ANN])
            for j, obj in enumerate(doc["objects"][:2]):
                pts = obj["polygon"]["ptList"]
                xs, ys = [p["x"] for p in pts], [p["y"] for p in pts]
                one = dict(obj, polygon={"ptList": [{"x": min(xs), "y": min(ys)}, {"x": max(xs), "y": max(ys)}]})
                img = os.path.join(imgdir, f"pic{i:02d}_{j}.png")
                open(img, "wb").write(b"png")
                recs.append({"source": img, "分类标签": obj["name"], NEW: json.dumps({"objects": [one]}, ensure_ascii=False),
                             ANN: base.loc[i, ANN], "width": doc["width"], "height": doc["height"]})
        sheet = pd.DataFrame(recs)
        sheet.loc[0, "source"] = None                      # 缺少source
        sheet.loc[1, "分类标签"] = np.nan                   # "nan" is not a class
        sheet.loc[2, NEW] = np.nan                         # NaN is truthy: no fallback, no box
        sheet.loc[3, NEW] = None                           # falls back to the original polygon column
        sheet.loc[4, "source"] = os.path.join(imgdir, "missing.jpg")     # no image on disk
        sheet.loc[5, "width"] = 0
        n = len(sheet)
        frames3 = {"train": sheet.iloc[: n * 6 // 10].reset_index(drop=True), "val": sheet.iloc[n * 6 // 10: n * 8 // 10].reset_index(drop=True),
                   "test": sheet.iloc[n * 8 // 10:].reset_index(drop=True)}
        run3 = run_reference_yolo(frames3, seed=7, class_order=["c5", "c3", "zzz"])
        nan_mark = lambda f: f.apply(lambda col: col.map(lambda v: "__NaN__" if isinstance(v, float) and v != v else v))  # noqa: E731
        frames3_json = {k: json.loads(nan_mark(v.assign(source=v["source"].map(lambda p: os.path.basename(p) if isinstance(p, str) else p)))
                                       .to_json(orient="split", force_ascii=False)) for k, v in frames3.items()}
    names = yaml.safe_load(run["data_yaml"])["names"]
    cls = {n: i for i, n in enumerate(names)}
    by_case = {}
    for fn, text in run["labels"]["train"].items():
        by_case[int(fn[4:7])] = text
    out = {"classes": names, "cases": {}}
    for k, (name, (cell, label, w, h)) in enumerate(YOLO_CASES.items()):
        boxes = ref_utils._extract_boxes_with_labels(cell)
        out["cases"][name] = {"json": cell, "label": label, "class_id": cls[label], "width": w, "height": h,
                              "boxes": [list(b) for b in boxes], "text": by_case.get(k)}
    reasons = {}
    for r in run["skipped"]:
        reasons[r["reason"]] = reasons.get(r["reason"], 0) + 1
    out["skipped_reasons"] = reasons
    out["stats"] = run["stats"]
    out["run"] = run                                       # the whole single-sheet run (file names, yaml, skipped order)
    out["run3"] = {"frames": frames3_json, "seed": 7, "class_order": ["c5", "c3", "zzz"], "result": run3}
    _dump("yolo_cases.json", out)


# ------------------------------------------------------------------------------- f3  merge
def merge_inputs():
    """file name -> text (written with utf-8-sig unless noted); sorted name order is the merge order"""
    t = synth.generate(23, seed=5, max_boxes=3)
    a = synth.to_frame(t)
    a["备注"] = ["x,1", 'q"q', "多行\n文本", "", "NA", "7"] * 3 + ["1.50", "007", "True", "nan", " pad "]
    a["n"] = range(len(a))
    b = synth.to_frame(synth.generate(9, seed=6, max_boxes=2))[[ANN, "source"]]          # other column order, no extras
    b.loc[3, ANN] = np.nan
    c = pd.DataFrame({"source": ["u1", "u2"], "k": [1.5, 2.0]})                            # no wide column: pandas path
    files = {
        "a_main.csv": a.to_csv(index=False),
        "b_other_order.csv": b.to_csv(index=False),
        "c_narrow.csv": c.to_csv(index=False),
        "d_crlf.csv": a.head(4).to_csv(index=False, lineterminator="\r\n"),
        "e_header_only.csv": "source," + ANN + "\n",
        "f_empty.csv": "",
        "g_ragged.csv": "source," + ANN + "\nonly_one_field\nu,{},extra\n",
        "h_has_source_file.csv": pd.DataFrame({"source": ["z"], ANN: ['{"objects": []}' + " " * 80], "source_file": ["old"]}).to_csv(index=False),
    }
    return files


def run_reference_merge(files, chunk_size):
    import contextlib
    import pathlib
    calls = []
    orig_glob = pathlib.Path.glob
    pathlib.Path.glob = lambda self, pat: iter(sorted(orig_glob(self, pat)))
    try:
        with tempfile.TemporaryDirectory() as d:
            folder = os.path.join(d, "in")
            os.makedirs(folder)
            for name, text in files.items():
                with open(os.path.join(folder, name), "w", encoding="utf-8-sig", newline="") as f:
                    f.write(text)
            out = os.path.join(d, "o", "merged.csv")
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ret = ref.merge_all_csv_in_folder(folder, out, "utf-8-sig", chunk_size, lambda *a: calls.append(list(a)))
            merged = open(out, "rb").read().decode("utf-8")          # keeps the BOM as \ufeff
            printed = buf.getvalue().replace(d, "<TMP>")
    finally:
        pathlib.Path.glob = orig_glob
    return {"chunk_size": chunk_size, "return": ret, "merged": merged, "calls": calls, "printed": printed}


def make_merge():
    files = merge_inputs()
    _dump("merge_case.json", {"files": files, "runs": [run_reference_merge(files, 7), run_reference_merge(files, 100000)]})


# ------------------------------------------------------------------------------- label_replace (between a4 and a5)
def run_reference_label_replace(df, mapping_df, **kwargs):
    """replace_labels_by_mapping with the Excel layer captured in memory; returns what it wrote and returned"""
    sheets = {}
    orig = (ref.pd.read_excel, pd.DataFrame.to_excel)

    def to_excel(self, target, *a, **k):
        sheets[os.path.basename(str(target))] = self.copy()

    ref.pd.read_excel = lambda *a, **k: mapping_df
    pd.DataFrame.to_excel = to_excel
    try:
        with tempfile.TemporaryDirectory() as d:
            inp, outp = os.path.join(d, "in.csv"), os.path.join(d, "o", "out.csv")
            df.to_csv(inp, index=False, encoding="utf-8-sig")
            try:
                res = ref.replace_labels_by_mapping(inp, os.path.join(d, "map.xlsx"), outp, diff_excel_path=os.path.join(d, "x", "diff.xlsx"),
                                                    unmatched_excel_path=os.path.join(d, "x", "unmatched.xlsx"), **kwargs)
            except Exception as e:  # noqa: BLE001
                return {"raises": type(e).__name__, "message": str(e)}
            return {"csv": _read_text(outp), "summary": res["summary"], "sample_diff": res["sample_diff"],
                    "diff_name": os.path.basename(str(res["diff"])), "unmatched_name": os.path.basename(str(res["unmatched"])),
                    "sheets": {k: _frame_records(v) for k, v in sheets.items()}}
    finally:
        ref.pd.read_excel, pd.DataFrame.to_excel = orig


LABEL_CELLS = [
    '{"width": 640, "height": 480, "objects": [{"name": "cat", "polygon": {"ptList": [{"x": 1, "y": 2}, {"x": 3.50, "y": 4e0}]}}, {"name": "dog,cat", "k": [1, 2.0, 1e22, 1e-7, -0.0]}]}',
    '{"objects":[{"name":"b,a"},{"name":" dog ； bird|dog "},{"name":"猫"},{"name":"\\u732b,wolf"}],"extra":{"a":null,"b":true,"c":"x\\ty\\"z\\\\"}}',
    '{"objects": [{"name": null}, {"polygon": {}}, {"name": ""}, 7, "s", [1], {"name": "kitty"}, {"name": "lion"}]}',
    '{"objects": {"name": "cat"}}',
    '{"objects": []}',
    '{"width": 3}',
    '{"objects": [',
    '',
    '{"objects": [{"name": "cat", "v": NaN, "w": Infinity, "u": -Infinity, "big": 123456789012345678901234567890, "f": 0.1, "g": 100.0, "h": 1E+2, "i": 1.5e300}]}',
    '  {"objects" : [ {"name" : "cat;cat;CAT"} ] , "objects2": 1, "name": "cat"}  ',
    '{"objects": [{"name": "5"}, {"name": "5.0"}, {"name": "nan"}, {"name": "x,y，z;w；v|u"}]}',
    '{"a": 1, "objects": [{"name": "dog"}], "a": 2}',
    '{"objects": [{"name": "tab\\there,cat", "s": "\\u0001\\u001f\\u007f\\u00e9\\ud83d\\ude00/\\/"}]}',
]


def make_label_replace():
    n = len(LABEL_CELLS)
    df = pd.DataFrame({"source": [f"s{i}" for i in range(n)], ANN: LABEL_CELLS, "other": list(range(n))})
    df.loc[n] = ["s_nan", np.nan, 99]
    both = df.copy()
    both[NEW] = list(reversed(LABEL_CELLS)) + ['{"objects": [{"name": "wolf"}]}']
    mapping = pd.DataFrame({"旧标签": ["cat", " dog ", "kitty", "bird", None, "nan", "ghost", "5", 5.0, "CAT", "猫", "lion", ""],
                            "新标签": ["feline", "canine", "feline", " avian ", "x", "y", None, "five", "five-float", "nan", "feline", " ", "z"],
                            "备注": list("abcdefghijklm")})
    out = {"mapping": _frame_records(mapping), "cases": {}}
    out["cases"]["one_column"] = {"input": _frame_records(df), "kwargs": {}, "result": run_reference_label_replace(df, mapping)}
    out["cases"]["both_columns_sample2"] = {"input": _frame_records(both), "kwargs": {"sample_size": 2},
                                            "result": run_reference_label_replace(both, mapping, sample_size=2)}
    out["cases"]["explicit_columns"] = {"input": _frame_records(df), "kwargs": {"old_col": "新标签", "new_col": "备注", "json_columns": [ANN, "missing"]},
                                        "result": run_reference_label_replace(df, mapping, old_col="新标签", new_col="备注", json_columns=[ANN, "missing"])}
    empty_map = pd.DataFrame({"a": [None], "b": ["x"]})
    out["cases"]["nothing_mapped"] = {"input": _frame_records(df.head(3)), "mapping": _frame_records(empty_map), "kwargs": {},
                                      "result": run_reference_label_replace(df.head(3), empty_map)}
    one_col = pd.DataFrame({"a": ["x"]})
    out["cases"]["mapping_one_column"] = {"input": _frame_records(df.head(2)), "mapping": _frame_records(one_col), "kwargs": {},
                                          "result": run_reference_label_replace(df.head(2), one_col)}
    # cells the reference does not survive (only JSONDecodeError is caught, :573-577)
    raising = {"list_document": "[1, 2]", "null_document": "null", "number_document": "12", "string_document": '"abc"',
               "int_name_changes": '{"objects": [{"name": 7}]}', "true_name": '{"objects": [{"name": true}]}',
               "list_name": '{"objects": [{"name": ["cat"]}]}', "float_name": '{"objects": [{"name": 2.5}]}',
               "zero_name_is_quiet": '{"objects": [{"name": 0}, {"name": false}, {"name": []}, {"name": {}}]}',
               "dict_name": '{"objects": [{"name": {"a": 1}}]}'}
    out["single_cells"] = {}
    for k, cell in raising.items():
        f = pd.DataFrame({"source": ["r0", "r1"], ANN: ['{"objects": [{"name": "cat"}]}', cell]})
        out["single_cells"][k] = {"cell": cell, "result": run_reference_label_replace(f, mapping)}
    nosrc = pd.DataFrame({ANN: ['{"objects": [{"name": "cat"}]}'], "n": [1]})
    out["cases"]["no_source_column"] = {"input": _frame_records(nosrc), "kwargs": {}, "result": run_reference_label_replace(nosrc, mapping)}
    numeric = pd.DataFrame({"source": ["a", "b"], ANN: [1.5, 2.5]})
    out["cases"]["numeric_annotation_column"] = {"input": _frame_records(numeric), "kwargs": {}, "result": run_reference_label_replace(numeric, mapping)}
    _dump("label_replace_case.json", out)


# ------------------------------------------------------------------------------- the two summaries either side of a5 / f4
def run_reference_unclassified(df, **kwargs):
    cap = _Capture()
    orig = (ref.pd.read_excel, ref.pd.ExcelWriter, pd.DataFrame.to_excel)

    def to_excel(self, target, sheet_name="Sheet1", index=True, **k):
        cap.sheets.setdefault(cap.current, {})[sheet_name] = self.copy()

    ref.pd.read_excel = lambda *a, **k: df.copy()
    ref.pd.ExcelWriter = cap.writer
    pd.DataFrame.to_excel = to_excel
    try:
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "unclassified.xlsx")
            open(src, "wb").close()
            try:
                out_path = ref.summarize_unclassified(src, os.path.join(d, "sum", "dir"), **kwargs)
            except Exception as e:  # noqa: BLE001
                return {"raises": type(e).__name__, "message": str(e)}
            return {"name": os.path.basename(str(out_path)), "parent_exists": os.path.isdir(os.path.dirname(str(out_path))),
                    "sheets": {s: _frame_records(f) for s, f in cap.sheets[os.path.basename(str(out_path))].items()}}
    finally:
        ref.pd.read_excel, ref.pd.ExcelWriter, pd.DataFrame.to_excel = orig


def make_summaries():
    out = {"unclassified": {}, "label_counts": {}}
    full = pd.DataFrame({"source": [f"u{i}" for i in range(12)],
                         "无法分类原因": ["标签foo未在规则中定义", "标签bar未在规则中定义", "标签foo未在规则中定义", "空数据", None, "JSON解析失败",
                                    "标签a,b未在规则中定义", "标签未在规则中定义", "没有可用标签", "标签foo未在规则中定义", "空数据", "x"],
                         "无法分类标签": ["foo", None, "foo, baz；foo", None, None, "", "a|b", None, " ", "qux", None, "foo"]})
    out["unclassified"]["full"] = {"input": _frame_records(full), "result": run_reference_unclassified(full)}
    no_labels = full.drop(columns=["无法分类标签"])
    out["unclassified"]["no_label_column"] = {"input": _frame_records(no_labels), "result": run_reference_unclassified(no_labels)}
    no_reason = pd.DataFrame({"source": ["a", "b"], "无法分类标签": ["k", None]})
    out["unclassified"]["no_reason_column"] = {"input": _frame_records(no_reason), "result": run_reference_unclassified(no_reason)}
    empty = pd.DataFrame({"source": [], "无法分类原因": []})
    out["unclassified"]["empty"] = {"input": _frame_records(empty), "result": run_reference_unclassified(empty)}
    try:
        ref.summarize_unclassified("/nonexistent/x.xlsx", "/tmp/never")
    except Exception as e:  # noqa: BLE001
        out["unclassified"]["missing_file"] = {"raises": type(e).__name__, "message": str(e)}

    # label files as generate_yolo_datasets_from_excels leaves them, plus the things the counter shrugs off
    tree = {
        "ds_a/data.yaml": "path: .\nnames:\n- cat\n- dog\n- bird\n",
        "ds_a/labels/train/i1.txt": "0 0.5 0.5 0.1 0.1\n1 0.2 0.2 0.1 0.1\n0 0.7 0.7 0.1 0.1",
        "ds_a/labels/train/i2.txt": "2 0.1 0.1 0.1 0.1\n\n   \n7 0.1 0.1 0.1 0.1\nx y z\n1.0 0.3 0.3 0.1 0.1\n-1 0.1 0.1 0.1 0.1\n",
        "ds_a/labels/train/empty.txt": "",
        "ds_a/labels/train/notes.md": "0 1 1 1 1",
        "ds_a/labels/val/v1.txt": "1 0.5 0.5 0.2 0.2",
        "ds_b/labels/test/t1.txt": "3 0.5 0.5 0.2 0.2\n3 0.1 0.1 0.1 0.1\n0 0.1 0.1 0.1 0.1",
        "ds_c/data.yaml": "names: [only\n",
        "ds_c/labels/train/c1.txt": "0 0.5 0.5 0.2 0.2",
        "ds_d/data.yaml": "names:\n  0: zero\n  1: one\n",
        "ds_d/labels/train/d1.txt": "1 0.5 0.5 0.2 0.2\n5 0.5 0.5 0.2 0.2",
        "ds_e/data.yaml": "nc: 2\n",
        "ds_e/labels/val/e1.txt": "1e0 0.5 0.5 0.2 0.2\nnan 0 0 0 0\ninf 0 0 0 0",
    }
    with tempfile.TemporaryDirectory() as d:
        for rel, text in tree.items():
            p = os.path.join(d, rel)
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "w", encoding="utf-8") as f:
                f.write(text)
        os.makedirs(os.path.join(d, "ds_empty"))
        dirs = [os.path.join(d, n) if n else n for n in ("ds_a", "ds_b", "", "ds_missing", "ds_c", "ds_d", "ds_e", "ds_empty")]
        calls = {}
        for name, arg in (("all", dirs), ("none", None), ("empty_list", []), ("only_missing", [os.path.join(d, "nope")])):
            try:
                stats, flat = ref.summarize_yolo_label_counts(arg)
                calls[name] = {"dirs": [os.path.basename(x) if x else x for x in (arg or [])], "arg_is_none": arg is None, "stats": stats,
                               "flat_columns": list(flat.columns), "flat_rows": json.loads(flat.to_json(orient="records", force_ascii=False))}
            except Exception as e:  # noqa: BLE001
                calls[name] = {"dirs": [os.path.basename(x) if x else x for x in (arg or [])], "raises": type(e).__name__, "message": str(e)}
    out["label_counts"] = {"tree": tree, "extra_dirs": ["ds_empty"], "calls": calls}
    _dump("summaries_case.json", out)


# ------------------------------------------------------------------------------- step "download": annotated images
def make_draw():
    """download_and_draw_annotations on images that are already in the download directory (no network here): the
    annotated PNGs it writes, byte for byte, plus the rows it gives up on (a connection refused at once, a broken file)"""
    import base64
    from PIL import Image

    def png(w, h, colour):
        buf = io.BytesIO()
        Image.new("RGB", (w, h), colour).save(buf, format="PNG")
        return buf.getvalue()

    images = {"a.png": png(96, 80, (10, 20, 30)), "b.png": png(64, 64, (200, 200, 200)), "image_3.jpg": png(50, 40, (0, 0, 0)),
              "broken.png": b"not an image", "d.png": png(40, 40, (255, 255, 255)), "e.png": png(32, 32, (1, 2, 3))}
    rows = [
        ("http://img.example/a.png", '{"objects": [{"name": "cat", "polygon": {"ptList": [{"x": 5, "y": 30}, {"x": 60, "y": 70}]}}, {"name": "多边形", "polygon": {"ptList": [{"x": 10, "y": 10}, {"x": 40.5, "y": 12}, {"x": 30, "y": 35.25}]}}]}',
         '{"objects": [{"name": "cat", "polygon": {"ptList": [{"x": 6, "y": 31}, {"x": 59, "y": 69}]}}]}'),
        ("http://img.example/dir/b.png", '{"objects": [{"polygon": {"ptList": [{"x": 1, "y": 25}, {"x": null, "y": 3}, {"x": 30, "y": 60}]}}, 7, {"name": "one", "polygon": {"ptList": [{"x": 1, "y": 2}]}}, {"name": "nopoly"}]}', np.nan),
        ("http://127.0.0.1:9/missing.png", '{"objects": []}', '{"objects": []}'),
        ("local-name-without-slash", '{"objects": [{"name": 5, "polygon": {"ptList": [{"x": 2, "y": 22}, {"x": 20, "y": 38}]}}]}', '{"objects": ['),
        ("http://img.example/broken.png", '{"objects": []}', '{"objects": []}'),
        ("http://img.example/d.png", '[1, 2]', '{"objects": [{"name": "late", "polygon": {"ptList": [{"x": 30, "y": 30}, {"x": 10, "y": 25}]}}, {"name": "bad", "polygon": {"ptList": "zz"}}, {"name": "after", "polygon": {"ptList": [{"x": 1, "y": 25}, {"x": 9, "y": 39}]}}]}'),
        ("http://img.example/e.png", '{"objects": [{"name": "x", "polygon": {"ptList": [{"x": 3, "y": 25}, {"x": 20, "y": 30}]}}]}', '{}'),
    ]
    df = pd.DataFrame(rows, columns=["source", ANN, NEW])
    out = {"input": _frame_records(df), "images": {k: base64.b64encode(v).decode() for k, v in images.items()}, "runs": {}}
    for name, max_images in (("all", None), ("first_three", 3)):
        with tempfile.TemporaryDirectory() as d:
            inp = os.path.join(d, "in.csv")
            df.to_csv(inp, index=False, encoding="utf-8-sig")
            os.makedirs(os.path.join(d, "out", "downloaded_images"))
            for k, v in images.items():
                with open(os.path.join(d, "out", "downloaded_images", k), "wb") as f:
                    f.write(v)
            buf = io.StringIO()
            import contextlib
            with contextlib.redirect_stdout(buf):
                ret = ref.download_and_draw_annotations(inp, os.path.join(d, "out"), None, None, max_images, 2)
            res_dir = os.path.join(d, "out", "annotated_images")
            out["runs"][name] = {"max_images": max_images, "returned_none": ret is None, "printed": buf.getvalue(),
                                 "annotated": {fn: base64.b64encode(open(os.path.join(res_dir, fn), "rb").read()).decode()
                                               for fn in sorted(os.listdir(res_dir))},
                                 "downloaded": sorted(os.listdir(os.path.join(d, "out", "downloaded_images")))}
    with tempfile.TemporaryDirectory() as d:       # the two early returns
        buf = io.StringIO()
        import contextlib
        with contextlib.redirect_stdout(buf):
            r1 = ref.download_and_draw_annotations(os.path.join(d, "nope.csv"), os.path.join(d, "o1"))
            pd.DataFrame({"source": ["a"]}).to_csv(os.path.join(d, "few.csv"), index=False, encoding="utf-8-sig")
            r2 = ref.download_and_draw_annotations(os.path.join(d, "few.csv"), os.path.join(d, "o2"))
        out["early"] = {"printed": buf.getvalue().replace(d, "<tmp>"), "returned": [r1 is None, r2 is None],
                        "dirs": sorted(os.listdir(os.path.join(d, "o1"))) + sorted(os.listdir(os.path.join(d, "o2")))}
    import PIL
    out["pillow"] = PIL.__version__
    _dump("draw_case.json", out)


if __name__ == "__main__":
    makers = {"replace": make_replace, "iou": make_iou, "dedup": make_dedup, "ref_filter": make_ref_filter, "perm": make_perm,
              "split": make_split, "e2e": make_e2e, "yolo": make_yolo, "merge": make_merge, "label_replace": make_label_replace,
              "summaries": make_summaries, "draw": make_draw, "chain": make_chain}
    for name in (sys.argv[1:] or list(makers)):
        makers[name]()
