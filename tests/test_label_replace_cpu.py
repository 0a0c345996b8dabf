"""The host steps either side of the device path that complete the processing page's imports (reference
ui/pages/processing.py:25-38): replace_labels_by_mapping (pipeline step label_replace), summarize_unclassified,
summarize_yolo_label_counts, overwrite_reference_with_result.  Product and oracle against what the reference produced
(tests/golden/label_replace_case.json, summaries_case.json; Excel layer captured in memory as in make_golden.py)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden
from deal_yolo_daya_amd.core import processor as P
from oracle import steps as osteps

GOLD = load_golden("label_replace_case.json")
SUMS = load_golden("summaries_case.json")


def _frame(rec):
    return pd.DataFrame(rec["data"], columns=rec["columns"], index=rec.get("index"))


def _records(f):
    return json.loads(f.to_json(orient="split", force_ascii=False))


class ExcelStub:
    """pandas' Excel layer, in memory (openpyxl is not installed): what make_golden.py does to the reference"""

    def __init__(self, monkeypatch, read=None):
        self.sheets, self.current = {}, None
        stub = self

        class Writer:
            def __init__(self, path, *a, **k):
                self.name = os.path.basename(str(path))

            def __enter__(self):
                stub.current = self.name
                return self

            def __exit__(self, *e):
                stub.current = None
                return False

        def to_excel(frame, target, sheet_name="Sheet1", index=True, **k):
            if isinstance(target, Writer):
                stub.sheets.setdefault(target.name, {})[sheet_name] = frame.copy()
            else:
                stub.sheets[os.path.basename(str(target))] = frame.copy()

        monkeypatch.setattr(pd, "read_excel", lambda *a, **k: read.copy())
        monkeypatch.setattr(pd, "ExcelWriter", Writer)
        monkeypatch.setattr(pd.DataFrame, "to_excel", to_excel)


def _run_product(tmp_path, monkeypatch, df, mapping, kwargs):
    stub = ExcelStub(monkeypatch, read=mapping)
    inp, outp = tmp_path / "in.csv", tmp_path / "o" / "out.csv"
    df.to_csv(inp, index=False, encoding="utf-8-sig")
    res = P.replace_labels_by_mapping(str(inp), str(tmp_path / "map.xlsx"), str(outp), diff_excel_path=str(tmp_path / "x" / "diff.xlsx"),
                                      unmatched_excel_path=str(tmp_path / "x" / "unmatched.xlsx"), **kwargs)
    return res, outp.read_bytes().decode("utf-8-sig"), stub.sheets


@pytest.mark.parametrize("name", sorted(GOLD["cases"]))
def test_label_replace_cases(tmp_path, monkeypatch, name):
    case = GOLD["cases"][name]
    mapping = _frame(case.get("mapping", GOLD["mapping"]))
    want = case["result"]
    df = _frame(case["input"])
    if "raises" in want:
        with pytest.raises(Exception) as ei:
            _run_product(tmp_path, monkeypatch, df, mapping, case["kwargs"])
        assert type(ei.value).__name__ == want["raises"] and str(ei.value) == want["message"]
        return
    res, text, sheets = _run_product(tmp_path, monkeypatch, df, mapping, case["kwargs"])
    assert text == want["csv"]
    assert res["summary"] == want["summary"]
    assert res["sample_diff"] == want["sample_diff"]
    assert os.path.basename(str(res["diff"])) == want["diff_name"] and os.path.basename(str(res["unmatched"])) == want["unmatched_name"]
    assert {k: _records(v) for k, v in sheets.items()} == want["sheets"]
    assert res["output_csv"] == tmp_path / "o" / "out.csv"


@pytest.mark.parametrize("name", sorted(GOLD["single_cells"]))
def test_label_replace_cells_the_reference_does_not_survive(tmp_path, monkeypatch, name):
    case = GOLD["single_cells"][name]
    df = pd.DataFrame({"source": ["r0", "r1"], P.ANNOTATION_COL: ['{"objects": [{"name": "cat"}]}', case["cell"]]})
    want = case["result"]
    mapping = _frame(GOLD["mapping"])
    if "raises" in want:
        with pytest.raises(Exception) as ei:
            _run_product(tmp_path, monkeypatch, df, mapping, {})
        assert type(ei.value).__name__ == want["raises"] and str(ei.value) == want["message"]
    else:
        res, text, sheets = _run_product(tmp_path, monkeypatch, df, mapping, {})
        assert text == want["csv"] and res["summary"] == want["summary"]


@pytest.mark.parametrize("name", sorted(GOLD["cases"]))
def test_oracle_label_replace_matches_the_reference(tmp_path, name):
    case = GOLD["cases"][name]
    mapping = _frame(case.get("mapping", GOLD["mapping"]))
    want = case["result"]
    inp = tmp_path / "in.csv"
    _frame(case["input"]).to_csv(inp, index=False, encoding="utf-8-sig")
    kw = dict(case["kwargs"])
    if "raises" in want:
        with pytest.raises(Exception) as ei:
            osteps.label_replace_csv(str(inp), mapping, str(tmp_path / "out.csv"), diff_excel_path="d", unmatched_excel_path="u", **kw)
        assert type(ei.value).__name__ == want["raises"] and str(ei.value) == want["message"]
        return
    res = osteps.label_replace_csv(str(inp), mapping, str(tmp_path / "out.csv"), diff_excel_path="d", unmatched_excel_path="u", **kw)
    assert (tmp_path / "out.csv").read_bytes().decode("utf-8-sig") == want["csv"]
    assert res["summary"] == want["summary"] and res["sample_diff"] == want["sample_diff"]
    assert _records(res["sheets"]["diff"]) == want["sheets"]["diff.xlsx"]
    assert _records(res["sheets"]["unmatched"]) == want["sheets"]["unmatched.xlsx"]


def test_label_replace_frame_twin_against_the_oracle_on_synthetic_rows():
    from deal_yolo_daya_amd import synth
    df = synth.to_frame(synth.generate(300, seed=11, max_boxes=6))
    rng = np.random.default_rng(3)
    label_map = {f"c{i}": f"g{int(rng.integers(0, 4))}" for i in range(0, 20, 2)}
    got, counters, diff, unmatched = P.replace_labels_frame(df, label_map)
    want, n, odiff, ounmatched = osteps.label_replace_frame(df, label_map)
    assert got.equals(want) and diff == odiff and list(unmatched.items()) == list(ounmatched.items())
    assert counters == {k: n[k] for k in counters}
    assert counters["replaced_objects"] > 100 and len(unmatched) > 3


@pytest.mark.parametrize("name", sorted(SUMS["unclassified"]))
def test_summarize_unclassified(tmp_path, monkeypatch, name):
    case = SUMS["unclassified"][name]
    if name == "missing_file":
        with pytest.raises(FileNotFoundError) as ei:
            P.summarize_unclassified("/nonexistent/x.xlsx", str(tmp_path / "never"))
        assert str(ei.value) == case["message"] and not (tmp_path / "never").exists()
        return
    df = _frame(case["input"])
    want = case["result"]
    stub = ExcelStub(monkeypatch, read=df)
    src = tmp_path / "unclassified.xlsx"
    src.write_bytes(b"")
    if "raises" in want:
        with pytest.raises(Exception) as ei:
            P.summarize_unclassified(str(src), str(tmp_path / "sum" / "dir"))
        assert type(ei.value).__name__ == want["raises"] and str(ei.value) == want["message"]
        with pytest.raises(Exception) as ei:
            osteps.unclassified_sheets(df)
        assert type(ei.value).__name__ == want["raises"]
        return
    out = P.summarize_unclassified(str(src), str(tmp_path / "sum" / "dir"), json_columns=["ignored"])
    assert out == tmp_path / "sum" / "dir" / want["name"] and out.parent.is_dir()
    assert {s: _records(f) for s, f in stub.sheets[want["name"]].items()} == want["sheets"]
    assert list(stub.sheets[want["name"]]) == ["reason_summary", "label_summary", "reason_label"]
    assert {s: _records(f) for s, f in osteps.unclassified_sheets(df).items()} == want["sheets"]


def _write_tree(root, spec):
    for rel, text in spec["tree"].items():
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text, encoding="utf-8")
    for d in spec["extra_dirs"]:
        (root / d).mkdir()


def _flat_set(rows):
    return sorted(json.dumps(r, ensure_ascii=False, sort_keys=True) for r in rows)


@pytest.mark.parametrize("name", sorted(SUMS["label_counts"]["calls"]))
@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_summarize_yolo_label_counts(tmp_path, name, impl):
    spec = SUMS["label_counts"]
    call = spec["calls"][name]
    _write_tree(tmp_path, spec)
    arg = None if call["arg_is_none"] else [str(tmp_path / d) if d else d for d in call["dirs"]]
    fn = P.summarize_yolo_label_counts if impl == "product" else osteps.yolo_label_counts
    stats, flat = fn(arg)
    assert stats == call["stats"] and list(stats) == list(call["stats"])
    assert list(flat.columns) == call["flat_columns"]
    # the reference lists the labels of a split in set order (hash-seed dependent): compare the rows as a set, the
    # (dataset, split) blocks in order
    got = json.loads(flat.to_json(orient="records", force_ascii=False)) if len(flat) else []
    assert _flat_set(got) == _flat_set(call["flat_rows"])
    assert [(r["数据集"], r["split"]) for r in got if True] == [(r["数据集"], r["split"]) for r in call["flat_rows"]]


def test_label_counts_read_back_what_the_label_step_writes(tmp_path):
    """label files in the exact form K7's text takes ("cid cx cy w h" lines joined by newlines, no trailing newline)"""
    (tmp_path / "ds" / "labels" / "train").mkdir(parents=True)
    (tmp_path / "ds" / "data.yaml").write_text("names:\n- a\n- b\n", encoding="utf-8")
    (tmp_path / "ds" / "labels" / "train" / "x.txt").write_text("0 0.500000 0.500000 0.100000 0.100000\n1 0.250000 0.250000 0.500000 0.500000\n1 0.1 0.1 0.1 0.1",
                                                              encoding="utf-8")
    stats, flat = P.summarize_yolo_label_counts([str(tmp_path / "ds")])
    assert stats["ds"]["train"] == {"total_images": 1, "label_counts": {"a": 1, "b": 1}, "box_counts": {"a": 1, "b": 2}}
    assert set(flat["占比%"]) == {"100.0%"}


def test_overwrite_reference_with_result(tmp_path):
    src, dst = tmp_path / "filtered.csv", tmp_path / "ref.csv"
    src.write_text("source\na\n", encoding="utf-8")
    dst.write_text("old", encoding="utf-8")
    os.utime(src, (1_600_000_000, 1_600_000_000))
    assert P.overwrite_reference_with_result(str(src), str(dst)) is None
    assert dst.read_text(encoding="utf-8") == "source\na\n" and int(dst.stat().st_mtime) == 1_600_000_000     # copy2 keeps the times
    with pytest.raises(FileNotFoundError) as ei:
        P.overwrite_reference_with_result(str(tmp_path / "nope.csv"), str(dst))
    assert str(ei.value) == f"结果文件不存在：{tmp_path / 'nope.csv'}"


def test_processing_page_imports_resolve():
    """every name the reference's processing page takes from core.processor (ui/pages/processing.py:25-38)"""
    for name in ("merge_all_csv_in_folder", "deduplicate_csv_by_source", "remove_duplicates_between_csv", "overwrite_reference_with_result",
                 "process_csv_replace_ptlist", "filter_by_box_count_and_iou", "replace_labels_by_mapping", "split_dataset_by_rules",
                 "summarize_unclassified", "generate_yolo_datasets_from_excels", "summarize_yolo_label_counts", "download_and_draw_annotations"):
        assert callable(getattr(P, name)), name


def _synthetic_two_column_table(n=400):
    from deal_yolo_daya_amd import synth
    df = synth.to_frame(synth.generate(n, seed=21, max_boxes=5))
    df[P.BBOX_COL] = df[P.ANNOTATION_COL].iloc[::-1].to_numpy()
    df.loc[3, P.BBOX_COL] = np.nan
    df.loc[5, P.ANNOTATION_COL] = '{"objects": ['                                           # decode error
    df.loc[7, P.ANNOTATION_COL] = '{"objects": [{"name": "c1", "name": "c2"}], "pad": "' + "x" * 80 + '"}'   # repeated key: CPython decides
    df.loc[9, P.BBOX_COL] = '{"objects": [{"na\\u006de": "c3,c4"}, {"name": "c4；c2"}], "pad": "' + "y" * 80 + '"}'
    df.loc[11, P.ANNOTATION_COL] = '{"objects": {"name": "c1"}, "pad": "' + "z" * 80 + '"}'
    df["n"] = np.arange(len(df))
    return df


@pytest.mark.parametrize("json_columns", [None, [P.ANNOTATION_COL], [P.ANNOTATION_COL, P.BBOX_COL, "absent"]])
def test_label_replace_csv_paths_agree(tmp_path, monkeypatch, json_columns):
    """native CSV + native relabeller, native relabeller behind pandas I/O, and CPython alone write the same bytes"""
    df = _synthetic_two_column_table()
    mapping = pd.DataFrame({"old": [f"c{i}" for i in range(0, 20, 3)], "new": ["g1", "g2", "g1", "c1", "g3", "g2", "g1"]})
    kwargs = {} if json_columns is None else {"json_columns": json_columns}
    runs = {}
    for mode, env in (("native", {}), ("native_json_only", {"DYD_NATIVE_CSV": "0"}), ("cpython", {"DYD_NATIVE_CSV": "0", "DYD_NATIVE_JSON": "0"})):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            d = tmp_path / mode
            d.mkdir()
            res, text, sheets = _run_product(d, m, df, mapping, kwargs)
            runs[mode] = (text, res["summary"], res["sample_diff"], {k: _records(v) for k, v in sheets.items()}, P.LAST_IO_PATH["label_replace"])
    assert runs["native"][4] == "native" and runs["native_json_only"][4] == "pandas" and runs["cpython"][4] == "pandas"
    assert runs["native"][:4] == runs["cpython"][:4] and runs["native_json_only"][:4] == runs["cpython"][:4]
    summary = runs["cpython"][1]
    assert summary["replaced_rows"] > 100 and summary["invalid_json_rows"] == 1 and summary["unmatched_labels"] > 5
    inp = tmp_path / "in.csv"
    df.to_csv(inp, index=False, encoding="utf-8-sig")
    want = osteps.label_replace_csv(str(inp), mapping, str(tmp_path / "oracle.csv"), diff_excel_path="d", unmatched_excel_path="u", **kwargs)
    assert (tmp_path / "oracle.csv").read_bytes().decode("utf-8-sig") == runs["native"][0] and want["summary"] == summary


def test_label_replace_csv_fast_path_raises_like_the_reference(tmp_path, monkeypatch):
    df = _synthetic_two_column_table(60)
    df.loc[20, P.ANNOTATION_COL] = '{"objects": [{"name": 7}], "pad": "' + "x" * 80 + '"}'
    mapping = pd.DataFrame({"old": ["c1"], "new": ["g1"]})
    with pytest.raises(TypeError) as ei:
        _run_product(tmp_path, monkeypatch, df, mapping, {})
    assert str(ei.value) == "sequence item 0: expected str instance, int found"
    assert not (tmp_path / "o" / "out.csv").exists()


DRAW = load_golden("draw_case.json")


@pytest.mark.parametrize("run", sorted(DRAW["runs"]))
def test_download_and_draw_annotations(tmp_path, capsys, run):
    """images already in the download directory (no network): the annotated files are the reference's, byte for byte"""
    import base64
    import PIL
    if PIL.__version__ != DRAW["pillow"]:
        pytest.skip("the fixture's bytes are Pillow %s's" % DRAW["pillow"])
    want = DRAW["runs"][run]
    df = _frame(DRAW["input"])
    inp = tmp_path / "in.csv"
    df.to_csv(inp, index=False, encoding="utf-8-sig")
    dl = tmp_path / "out" / "downloaded_images"
    dl.mkdir(parents=True)
    for name, b64 in DRAW["images"].items():
        (dl / name).write_bytes(base64.b64decode(b64))
    assert P.download_and_draw_annotations(str(inp), str(tmp_path / "out"), None, None, want["max_images"], 2) is None
    assert capsys.readouterr().out == want["printed"]
    res = tmp_path / "out" / "annotated_images"
    assert sorted(p.name for p in res.iterdir()) == sorted(want["annotated"])
    for name, b64 in want["annotated"].items():
        assert (res / name).read_bytes() == base64.b64decode(b64), name
    assert sorted(p.name for p in dl.iterdir()) == want["downloaded"]


def test_download_and_draw_annotations_early_returns(tmp_path, capsys):
    want = DRAW["early"]
    assert P.download_and_draw_annotations(str(tmp_path / "nope.csv"), str(tmp_path / "o1")) is None
    pd.DataFrame({"source": ["a"]}).to_csv(tmp_path / "few.csv", index=False, encoding="utf-8-sig")
    assert P.download_and_draw_annotations(str(tmp_path / "few.csv"), str(tmp_path / "o2")) is None
    assert capsys.readouterr().out.replace(str(tmp_path), "<tmp>") == want["printed"]
    assert sorted(p.name for p in (tmp_path / "o1").iterdir()) + sorted(p.name for p in (tmp_path / "o2").iterdir()) == want["dirs"]
