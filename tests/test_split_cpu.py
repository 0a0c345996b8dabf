"""The split step's host side with the threaded column builders switched on (they normally start at 65536 records; the CPU
port of the reference, oracle/steps.py split_frames = processor.py:654-831, walks 150 rows a second, so the tests lower the
threshold instead of growing the table), the oracle standing in for K8 + K6."""
import json

import numpy as np
import pandas as pd
import pytest

from deal_yolo_daya_amd import synth
from deal_yolo_daya_amd.core import processor as P
from oracle import steps as osteps


@pytest.fixture(autouse=True, params=["one call", "row batches"])
def threaded_builders(request, monkeypatch):
    """every test twice: the table expanded by one native call, and in row batches with the records' str objects allocated
    while later batches are parsed (processor.SPLIT_BATCH_ROWS is 40 000 rows in production)"""
    from deal_yolo_daya_amd import pycells
    pycells.set_min_threaded(64)
    if request.param == "row batches":
        monkeypatch.setattr(P, "SPLIT_BATCH_ROWS", 90)
    yield
    pycells.set_min_threaded()


def _table(n_rows, seed, backend):
    df = synth.to_frame(synth.generate(n_rows, seed=seed))
    kept, _ = P.replace_ptlist_frame(df, backend=backend)
    return kept


def _same_frames(got, exp):
    assert list(got["categories"]) == list(exp["categories"])
    for cat in exp["categories"]:
        for a, b in zip(got["categories"][cat], exp["categories"][cat]):
            pd.testing.assert_frame_equal(a, b)
            assert a.equals(b)
    assert got["unclassified"].equals(exp["unclassified"]) and got["split_counts"].equals(exp["split_counts"])
    assert got["category_counts"] == exp["category_counts"]


def test_split_frames_equal_the_cpu_port_beyond_the_thread_threshold(oracle_backend):
    table = _table(520, 11, oracle_backend)
    rules = synth.rules()
    got = P.split_frames(table, rules, backend=oracle_backend, stats=(st := {}))
    assert st["records"] > 5000 and st["fast_cells"] == len(table)
    assert st["expand_batches"] == (len(table) // 90 if P.SPLIT_BATCH_ROWS == 90 else 1)
    _same_frames(got, osteps.split_frames(table, rules))
    # ratios that do not sum to one, another seed
    got = P.split_frames(table, rules, None, 6, 3, 1, random_seed=7, backend=oracle_backend)
    _same_frames(got, osteps.split_frames(table, rules, None, 6, 3, 1, random_seed=7))


def test_split_frames_with_cells_of_every_kind(oracle_backend):
    """rows the native lane hands to the exact walker, rows only CPython decides, error rows, rows without a usable cell,
    labels outside the rules, Chinese labels and separators — mixed into a table large enough for the threaded builders"""
    table = _table(700, 12, oracle_backend).reset_index(drop=True)
    cells = table[P.BBOX_COL].tolist()
    odd = [
        '{"objects": [{"name": "c1，c2;c3|c99 ", "polygon": {"ptList": []}}, {"name": null}, {"nam": 1}, 7], "width": 1e5}',
        '{"objects": [{"name": 7}], "a": 1}',                        # numeric name: the Python path
        '{"objects": [{"name": "c1"}], "objects": []}',             # repeated key
        '{"objects": {"name": "c1"}}', '{"objects": []}', '[1, 2]', '{"objects": [', "", None, float("nan"), 17,
        '{"k": "\\u4e2d\\u6587", "objects": [{"name": "\\u7532,c2", "x": "\\ud83d\\ude00"}]}',
        '{"objects": [{"name": "甲；乙 ，c3", "v": [1.0, 2.50, -0.0, 1e-7, 123456789012345678]}], "z": {"q": null}}',
        '{"objects": [{"name": "c4", "name2": "x"}, {"name": ""}, {"name": false}, {"name": []}]}',
    ]
    rng = np.random.default_rng(5)
    for i, k in enumerate(rng.choice(len(cells), size=8 * len(odd), replace=False).tolist()):
        cells[k] = odd[i % len(odd)]
    table[P.BBOX_COL] = pd.Series(cells, dtype=object)
    table.loc[rng.choice(len(table), 20, replace=False), P.ANNOTATION_COL] = None
    rules = dict(synth.rules(), **{"甲": "catC", "c3": "catC"})
    got = P.split_frames(table, rules, backend=oracle_backend, stats=(st := {}))
    assert st["records"] > 5000 and 0 < st["fast_cells"] < len(table)
    exp = osteps.split_frames(table, rules)
    _same_frames(got, exp)
    assert "catC" in got["categories"] and len(got["unclassified"]) > 500


def test_arrow_text_columns_hold_the_same_values(oracle_backend):
    pytest.importorskip("pyarrow")
    table = _table(600, 13, oracle_backend)
    rules = synth.rules()
    a = P.split_frames(table, rules, backend=oracle_backend)
    b = P.split_frames(table, rules, backend=oracle_backend, text_dtype="arrow")
    for cat in a["categories"]:
        for fa, fb in zip(a["categories"][cat], b["categories"][cat]):
            assert str(fb[P.BBOX_COL].dtype) == "string" and fa[P.BBOX_COL].dtype == object
            assert fa[P.BBOX_COL].tolist() == fb[P.BBOX_COL].tolist() and fa[P.ANNOTATION_COL].tolist() == fb[P.ANNOTATION_COL].tolist()
            pd.testing.assert_frame_equal(fa.drop(columns=[P.BBOX_COL, P.ANNOTATION_COL]), fb.drop(columns=[P.BBOX_COL, P.ANNOTATION_COL]))
    with pytest.raises(ValueError):
        P.split_frames(table, rules, backend=oracle_backend, text_dtype="bytes")


def test_split_frames_keeps_other_columns_dtypes_and_row_labels(oracle_backend):
    table = _table(300, 14, oracle_backend)
    table["score"] = np.linspace(0, 1, len(table))
    table["flag"] = np.arange(len(table)) % 2 == 0
    table["分类标签"] = "old"                                        # a column named like a new one is overwritten in place
    table.index = np.arange(len(table)) * 3 + 1000                  # row labels survive in the unclassified sheet only
    rules = synth.rules()
    got, exp = P.split_frames(table, rules, backend=oracle_backend), osteps.split_frames(table, rules)
    _same_frames(got, exp)
    frame = got["categories"]["catA"][0]
    assert frame["score"].dtype == np.float64 and frame["flag"].dtype == bool and list(frame.columns) == list(exp["categories"]["catA"][0].columns)
    first = json.loads(frame[P.BBOX_COL].iloc[0])
    assert len(first["objects"]) == 1 and first["objects"][0]["name"] == frame["分类标签"].iloc[0]


def test_the_tablewise_checker_is_the_cpu_port(oracle_backend):
    """tests/helpers.split_expected_tablewise (used by the GPU suite at 100 k rows, where the port itself would run for minutes)
    gives exactly what oracle.steps.split_frames gives"""
    from helpers import split_expected_tablewise
    table = _table(400, 15, oracle_backend).reset_index(drop=True)
    cells = table[P.BBOX_COL].tolist()
    for k, odd in zip(range(0, 400, 23), ['{"objects": [{"name": 7}]}', "", None, '{"objects": []}', '[1]', '{"objects": [{"nam": 1}, {"name": "c1，c99"}]}'] * 3):
        cells[k] = odd
    table[P.BBOX_COL] = pd.Series(cells, dtype=object)
    rules = synth.rules()
    _same_frames(split_expected_tablewise(table, rules), osteps.split_frames(table, rules))
    _same_frames(split_expected_tablewise(table, rules, None, 6, 3, 1, 7), osteps.split_frames(table, rules, None, 6, 3, 1, random_seed=7))


def test_split_frames_on_degenerate_tables(oracle_backend):
    """no rows, only error rows, nothing classified, no JSON column, a single record, no source column, the second JSON column
    standing in where the first is empty (:713-718): frames, side tables and counts of the CPU port"""
    rules = synth.rules()
    cases = {
        "empty": pd.DataFrame({"source": pd.Series([], dtype=object), P.BBOX_COL: pd.Series([], dtype=object)}),
        "all errors": pd.DataFrame({"source": ["a", "b", "c"], P.BBOX_COL: [None, "{", '{"objects": 5}']}),
        "nothing classified": pd.DataFrame({"source": ["a", "b"], P.BBOX_COL: ['{"objects": [{"name": "zz"}]}', '{"objects": [{"nam": 1}]}']}),
        "no json column": pd.DataFrame({"source": ["a"], "other": [1]}),
        "one record": pd.DataFrame({"source": ["a"], P.BBOX_COL: ['{"objects": [{"name": "c1"}], "w": 1}'], "n": [7]}),
        "no source": pd.DataFrame({P.ANNOTATION_COL: ['{"objects": [{"name": "c1;c11"}]}', '{"objects": [{"name": "c12"}]}'] * 3}),
        "second column": pd.DataFrame({"source": list("abc"), P.BBOX_COL: ["", None, '{"objects": [{"name": "c2"}]}'],
                                       P.ANNOTATION_COL: ['{"objects": [{"name": "c3"}]}', '{"objects": [{"name": "c13"}]}', '{"objects": [{"name": "c4"}]}']}),
    }
    for name, df in cases.items():
        got, exp = P.split_frames(df, rules, backend=oracle_backend), osteps.split_frames(df, rules)
        assert list(got["categories"]) == list(exp["categories"]) and got["category_counts"] == exp["category_counts"], name
        for c in exp["categories"]:
            for a, b in zip(got["categories"][c], exp["categories"][c]):
                pd.testing.assert_frame_equal(a, b)
        for key in ("unclassified", "split_counts"):
            if len(exp[key]) or len(got[key]):
                pd.testing.assert_frame_equal(got[key], exp[key])
