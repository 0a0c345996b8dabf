"""Pin the CPU oracle (oracle/steps.py, oracle/dyd_oracle.c) against the golden vectors that
tests/golden/make_golden.py produced by running the reference itself."""
import io
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import golden_csv_text, load_golden
from helpers import frame_records, read_text, write_csv_text
from oracle import lib as olib
from oracle import steps as osteps

ANN, NEW = osteps.ANN_COL, osteps.NEW_COL
EXC = {"TypeError": TypeError, "AttributeError": AttributeError}


# ------------------------------------------------------------------------------ a3
def test_replace_value_cases():
    g = load_golden("replace_cases.json")
    for name, case in g["value_cases"].items():
        assert osteps.replace_cell(case["in"]) == case["out"], name
        w, h = osteps.width_height_of_cell(case["in"])
        for got, want in ((w, case["width"]), (h, case["height"])):
            if want is None:
                assert got is None or (name == "width_height_types"), name
            else:
                assert float(got) == float(want), name


def test_replace_raising_cases():
    g = load_golden("replace_cases.json")
    for name, case in g["raising_cases"].items():
        assert case["raises"] is not None, name
        with pytest.raises(EXC[case["raises"]]):
            osteps.replace_cell(case["in"])


def test_replace_csv_bytes(tmp_path):
    g = load_golden("replace_cases.json")
    inp, out, exc = (str(tmp_path / n) for n in ("in.csv", "out.csv", "exc.csv"))
    write_csv_text(inp, g["input_csv"])
    res = osteps.replace_csv(inp, out, exc)
    assert res["filtered_rows"] == g["result"]["filtered_rows"]
    assert res["excluded_rows"] == g["result"]["excluded_rows"]
    assert read_text(out) == g["output_csv"]
    assert read_text(exc) == g["excluded_csv"]


# ------------------------------------------------------------------------------ a4
def test_iou_cases_python_port():
    g = load_golden("iou_cases.json")
    for run in g["runs"]:
        got = [int(osteps.row_is_high(osteps.boxes_of_cell(c), run["min_boxes"], run["thr"])) for c in g["cells"]]
        assert got == run["high"], (run["min_boxes"], run["thr"])
    for name, case in g["raising"].items():
        with pytest.raises(EXC[case["raises"]]):
            osteps.row_is_high(osteps.boxes_of_cell(case["in"]), 2, 0.98)


def test_iou_csv_bytes(tmp_path):
    g = load_golden("iou_cases.json")
    run = next(r for r in g["runs"] if "input_csv" in r)
    inp, hi, lo = (str(tmp_path / n) for n in ("in.csv", "hi.csv", "lo.csv"))
    write_csv_text(inp, run["input_csv"])
    osteps.iou_filter_csv(inp, hi, lo, run["min_boxes"], run["thr"])
    assert read_text(hi) == run["high_csv"]
    assert read_text(lo) == run["other_csv"]


def _numeric_rows(cells):
    """rows of the golden IoU matrix whose boxes are plain finite-or-not doubles -> flat arrays"""
    from deal_yolo_daya_amd import flatten
    batch = flatten.flatten_boxes(cells)
    return batch


def test_iou_cases_c_oracle():
    """dyd_oracle.c (f64) reproduces the reference's HIGH mask on every f64-representable row."""
    g = load_golden("iou_cases.json")
    batch = _numeric_rows(g["cells"])
    for run in g["runs"]:
        got = olib.iou_any_ge(batch.box4, batch.row_off, run["min_boxes"], run["thr"])
        for ri, want in enumerate(run["high"]):
            if ri in batch.host_rows:
                continue
            assert int(got[ri]) == want, (g["names"][ri], run["min_boxes"], run["thr"])
    assert set(g["names"][r] for r in batch.host_rows) == {"big_ints"}


# ------------------------------------------------------------------------------ a1 / a2
def test_dedup_cases(tmp_path):
    g = load_golden("dedup_cases.json")
    for name, case in g.items():
        inp = str(tmp_path / f"{name}.csv")
        write_csv_text(inp, case["input_csv"])
        for keep_s, want in case["keep"].items():
            keep = False if keep_s == "False" else keep_s
            out = str(tmp_path / "o.csv")
            res = osteps.dedup_csv(inp, out, keep=keep)
            assert len(res) == want["rows"], (name, keep)
            assert read_text(out) == want["output_csv"], (name, keep)


def test_ref_filter_cases(tmp_path):
    g = load_golden("ref_filter_cases.json")
    for name, case in g.items():
        m, r, o = (str(tmp_path / n) for n in ("m.csv", "r.csv", "o.csv"))
        write_csv_text(m, case["main_csv"])
        write_csv_text(r, case["ref_csv"])
        res = osteps.ref_filter_csv(m, r, o, compare_col=case["compare_col"])
        assert len(res) == case["rows"], name
        assert read_text(o) == case["output_csv"], name


def test_hash_known_answers():
    """MurmurHash3 x64_128 seed 0 known answers (published test values of the algorithm)."""
    def h(b):
        data = np.frombuffer(b, np.uint8)
        return olib.hash128(data, np.array([0, len(b)], np.int64))[0]
    assert tuple(h(b"")) == (0, 0)
    assert tuple(int(v) for v in h(b"hello")) == (0xcbd8a7b341bd9b02, 0x5b1e906a48ae1d19)
    a = h(b"The quick brown fox jumps over the lazy dog")
    assert tuple(int(v) for v in a) == (0xe34bbc7bbc071b6c, 0x7a433ca9c49a9347)


# ------------------------------------------------------------------------------ a5
def test_permutation_vectors():
    g = load_golden("perm_cases.json")
    for case in g["perms"]:
        got = olib.mt19937_permutation(case["seed"], case["n"])
        assert got.tolist() == case["order"], (case["seed"], case["n"])
        # numpy's legacy generator is the third-party code the reference reaches through pandas
        assert got.tolist() == np.random.RandomState(case["seed"]).permutation(case["n"]).tolist()


def test_cut_sizes():
    g = load_golden("perm_cases.json")
    for c in g["cuts"]:
        tr, va, te = c["ratios"]
        s = tr + va + te
        assert int(c["n"] * (tr / s)) == c["n_train"] and int(c["n"] * (va / s)) == c["n_val"]


def _frames_equal(got: pd.DataFrame, want_records, ctx):
    want = pd.DataFrame(want_records["data"], columns=want_records["columns"])
    assert list(got.columns) == list(want.columns), ctx
    assert len(got) == len(want), ctx
    g = json.loads(got.reset_index(drop=True).to_json(orient="split", force_ascii=False))["data"]
    assert g == want_records["data"], ctx


def _split_with_oracle(golden_name, df=None):
    g = load_golden(golden_name)
    if df is None:
        df = pd.DataFrame(g["input"]["data"], columns=g["input"]["columns"])
    rules = pd.DataFrame(g["rules"]["data"], columns=g["rules"]["columns"])
    res = osteps.split_frames(df, osteps.rules_to_map(rules), random_seed=g.get("seed", 42))
    return g, res


def test_split_case():
    g, res = _split_with_oracle("split_case.json")
    assert res["category_counts"] == g["summary"]["category_counts"]
    assert len(res["unclassified"]) == g["summary"]["unclassified"]
    for cat, (tr, va, te) in res["categories"].items():
        sheets = g["sheets"][f"{cat}.xlsx"]
        for name, frame in (("train", tr), ("val", va), ("test", te)):
            _frames_equal(frame, sheets[name], (cat, name))
    _frames_equal(res["unclassified"], g["sheets"]["unclassified.xlsx"]["Sheet1"], "unclassified")
    _frames_equal(res["split_counts"], g["sheets"]["split_counts.xlsx"]["Sheet1"], "split_counts")


# ------------------------------------------------------------------------------ e2e
def test_e2e_chain(tmp_path):
    P = lambda n: str(tmp_path / n)  # noqa: E731
    for n in ("merged", "ref"):
        write_csv_text(P(n + ".csv"), golden_csv_text(f"e2e_{n}.csv.gz"))
    osteps.dedup_csv(P("merged.csv"), P("dedup.csv"))
    osteps.ref_filter_csv(P("dedup.csv"), P("ref.csv"), P("filtered.csv"))
    osteps.replace_csv(P("filtered.csv"), P("processed.csv"), P("excluded.csv"))
    osteps.iou_filter_csv(P("processed.csv"), P("high.csv"), P("other.csv"), 2, 0.98)
    for n in ("dedup", "filtered", "processed", "excluded", "high", "other"):
        assert read_text(P(n + ".csv")) == golden_csv_text(f"e2e_{n}.csv.gz"), n
    other = pd.read_csv(P("other.csv"), encoding="utf-8-sig")
    g, res = _split_with_oracle("e2e_split.json", other)
    assert res["category_counts"] == g["summary"]["category_counts"]
    for cat, (tr, va, te) in res["categories"].items():
        for name, frame in (("train", tr), ("val", va), ("test", te)):
            want = g["sheets"][f"{cat}.xlsx"][name]
            _frames_equal(frame[want["columns"]], want, (cat, name))


# ------------------------------------------------------------------ f4  YOLO label lines
def _same_value(a, b):
    if isinstance(a, float) and isinstance(b, float):
        return (a != a and b != b) or (a == b and np.signbit(a) == np.signbit(b))
    return type(a) is type(b) and a == b


def test_yolo_extract_matches_reference():
    g = load_golden("yolo_cases.json")
    for name, c in g["cases"].items():
        got = osteps.extract_boxes_with_labels(c["json"])
        assert len(got) == len(c["boxes"]), name
        for a, b in zip(got, c["boxes"]):
            assert all(_same_value(x, y) for x, y in zip(a, b)), (name, a, b)


def test_yolo_label_text_matches_reference():
    g = load_golden("yolo_cases.json")
    reasons = {}
    for name, c in g["cases"].items():
        text, why = osteps.yolo_row_text(c["json"], c["label"], c["class_id"], c["width"], c["height"])
        assert text == c["text"], name
        if why:
            reasons[why] = reasons.get(why, 0) + 1
    filler = 12                                             # filler rows of the fixture: no objects -> no matching box
    reasons["无匹配标签框"] += filler
    assert reasons == g["skipped_reasons"]


# ------------------------------------------------------------------ f3  merge
@pytest.mark.parametrize("run_idx", [0, 1])
def test_merge_port_matches_reference(tmp_path, monkeypatch, capsys, run_idx):
    import pathlib
    orig = pathlib.Path.glob
    monkeypatch.setattr(pathlib.Path, "glob", lambda self, pat: iter(sorted(orig(self, pat))))
    g = load_golden("merge_case.json")
    folder = tmp_path / "in"
    folder.mkdir()
    for name, text in g["files"].items():
        with open(folder / name, "w", encoding="utf-8-sig", newline="") as f:
            f.write(text)
    want = g["runs"][run_idx]
    calls = []
    out = tmp_path / "o" / "merged.csv"
    ret = osteps.merge_folder(str(folder), str(out), "utf-8-sig", want["chunk_size"], lambda *a: calls.append(list(a)))
    assert ret == want["return"] and calls == want["calls"]
    assert open(out, "rb").read().decode("utf-8") == want["merged"]
    assert capsys.readouterr().out.replace(str(tmp_path), "<TMP>") == want["printed"]
