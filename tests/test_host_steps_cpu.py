"""Host logic of the drop-in step functions (flatten / emit / CSV contract / error behaviour),
driven on CPU with the oracle standing in for the device stage, against the golden outputs of
the reference.  The same golden checks run against the real HIP kernels in test_gpu_parity.py."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import golden_csv_text, load_golden
from helpers import read_text, write_csv_text
from deal_yolo_daya_amd import flatten
from deal_yolo_daya_amd.core import processor as P

EXC = {"TypeError": TypeError, "AttributeError": AttributeError}


def run_replace_golden(backend, tmp_path):
    g = load_golden("replace_cases.json")
    inp, out, exc = (str(tmp_path / n) for n in ("in.csv", "out.csv", "exc.csv"))
    write_csv_text(inp, g["input_csv"])
    res = P.process_csv_replace_ptlist(inp, out, exc, backend=backend)
    assert res == {"filtered_rows": g["result"]["filtered_rows"], "excluded_rows": g["result"]["excluded_rows"],
                   "excluded_output": exc}
    assert read_text(out) == g["output_csv"]
    assert read_text(exc) == g["excluded_csv"]
    for name, case in g["value_cases"].items():
        texts, _, _ = P.replace_ptlist_cells([case["in"]], backend)
        assert texts[0] == case["out"], name
    for name, case in g["raising_cases"].items():
        with pytest.raises(EXC[case["raises"]]):
            P.replace_ptlist_cells(['{"objects": []}', case["in"]], backend)


def run_iou_golden(backend, tmp_path):
    g = load_golden("iou_cases.json")
    for run in g["runs"]:
        stats = {}
        mask = P.iou_high_mask(g["cells"], run["min_boxes"], run["thr"], backend, stats)
        assert mask.astype(int).tolist() == run["high"], (run["min_boxes"], run["thr"])
        assert stats["host_rows"] == 1            # only the >2^25 integer row is resolved on the host
        if "input_csv" in run:
            inp, hi, lo = (str(tmp_path / n) for n in ("in.csv", "hi.csv", "lo.csv"))
            write_csv_text(inp, run["input_csv"])
            assert P.filter_by_box_count_and_iou(inp, hi, lo, run["min_boxes"], run["thr"], backend=backend) is None
            assert read_text(hi) == run["high_csv"]
            assert read_text(lo) == run["other_csv"]
    for name, case in g["raising"].items():
        with pytest.raises(EXC[case["raises"]]):
            P.iou_high_mask([case["in"]], 2, 0.98, backend)


def run_dedup_golden(backend, tmp_path):
    g = load_golden("dedup_cases.json")
    for name, case in g.items():
        inp = str(tmp_path / f"{name}.csv")
        write_csv_text(inp, case["input_csv"])
        for keep_s, want in case["keep"].items():
            keep = False if keep_s == "False" else keep_s
            out = str(tmp_path / "o.csv")
            res = P.deduplicate_csv_by_source(inp, out, keep=keep, verbose=False, backend=backend)
            assert len(res) == want["rows"], (name, keep)
            assert read_text(out) == want["output_csv"], (name, keep)
            # the frame twin agrees with pandas itself (the reference's call at processor.py:140)
            df = pd.read_csv(inp, encoding="utf-8-sig", parse_dates=False)
            pd.testing.assert_frame_equal(P.dedup_frame(df, keep, backend),
                                          df.drop_duplicates(subset=["source"], keep=keep, ignore_index=True))


def run_ref_filter_golden(backend, tmp_path):
    g = load_golden("ref_filter_cases.json")
    for name, case in g.items():
        m, r, o = (str(tmp_path / n) for n in ("m.csv", "r.csv", "o.csv"))
        write_csv_text(m, case["main_csv"])
        write_csv_text(r, case["ref_csv"])
        res = P.remove_duplicates_between_csv(m, r, o, compare_col=case["compare_col"], verbose=False, backend=backend)
        assert len(res) == case["rows"], name
        assert read_text(o) == case["output_csv"], name


def _frames_equal(got, want_records, ctx):
    assert list(got.columns) == list(want_records["columns"]), ctx
    data = json.loads(got.reset_index(drop=True).to_json(orient="split", force_ascii=False))["data"]
    assert data == want_records["data"], ctx


def run_split_golden(backend):
    g = load_golden("split_case.json")
    df = pd.DataFrame(g["input"]["data"], columns=g["input"]["columns"])
    rules = pd.DataFrame(g["rules"]["data"], columns=g["rules"]["columns"])
    res = P.split_frames(df, P.rules_to_label_map(rules), random_seed=g["seed"], backend=backend)
    assert res["category_counts"] == g["summary"]["category_counts"]
    assert list(res["categories"]) == [f[:-5] for f in g["category_files"]]       # first-appearance order
    for cat, frames in res["categories"].items():
        for name, frame in zip(("train", "val", "test"), frames):
            _frames_equal(frame, g["sheets"][f"{cat}.xlsx"][name], (cat, name))
    _frames_equal(res["unclassified"], g["sheets"]["unclassified.xlsx"]["Sheet1"], "unclassified")
    _frames_equal(res["split_counts"], g["sheets"]["split_counts.xlsx"]["Sheet1"], "split_counts")
    t = g["two_column"]                                  # rule_mode="two_column", ratios 6:3:1, seed 7
    rules = pd.DataFrame(t["rules"]["data"], columns=t["rules"]["columns"])
    res = P.split_frames(df, P.rules_to_label_map(rules, "two_column", t["label_col"], t["category_col"]), None, *t["ratios"],
                         random_seed=t["seed"], backend=backend)
    assert res["category_counts"] == t["summary"]["category_counts"]
    assert list(res["categories"]) == [f[:-5] for f in t["category_files"]]
    for cat, frames in res["categories"].items():
        for name, frame in zip(("train", "val", "test"), frames):
            _frames_equal(frame, t["sheets"][f"{cat}.xlsx"][name], (cat, name, "two_column"))
    _frames_equal(res["unclassified"], t["sheets"]["unclassified.xlsx"]["Sheet1"], "unclassified two_column")
    _frames_equal(res["split_counts"], t["sheets"]["split_counts.xlsx"]["Sheet1"], "split_counts two_column")


def run_e2e_golden(backend, tmp_path):
    Q = lambda n: str(tmp_path / n)  # noqa: E731
    for n in ("merged", "ref"):
        write_csv_text(Q(n + ".csv"), golden_csv_text(f"e2e_{n}.csv.gz"))
    P.deduplicate_csv_by_source(Q("merged.csv"), Q("dedup.csv"), verbose=False, backend=backend)
    P.remove_duplicates_between_csv(Q("dedup.csv"), Q("ref.csv"), Q("filtered.csv"), verbose=False, backend=backend)
    P.process_csv_replace_ptlist(Q("filtered.csv"), Q("processed.csv"), Q("excluded.csv"), backend=backend)
    P.filter_by_box_count_and_iou(Q("processed.csv"), Q("high.csv"), Q("other.csv"), 2, 0.98, backend=backend)
    for n in ("dedup", "filtered", "processed", "excluded", "high", "other"):
        assert read_text(Q(n + ".csv")) == golden_csv_text(f"e2e_{n}.csv.gz"), n
    g = load_golden("e2e_split.json")
    other = pd.read_csv(Q("other.csv"), encoding="utf-8-sig")
    rules = pd.DataFrame(g["rules"]["data"], columns=g["rules"]["columns"])
    res = P.split_frames(other, P.rules_to_label_map(rules), backend=backend)
    assert res["category_counts"] == g["summary"]["category_counts"]
    for cat, frames in res["categories"].items():
        for name, frame in zip(("train", "val", "test"), frames):
            want = g["sheets"][f"{cat}.xlsx"][name]
            _frames_equal(frame[want["columns"]], want, (cat, name))


def run_chain_golden(backend, tmp_path):
    """replace -> IoU in ONE pass (fused K1+K2 behind process_csv_replace_and_filter / replace_and_filter_frame) against what the
    reference's two steps wrote in sequence: tests/golden/chain_cases.json ([A, null, A] is not HIGH, [A, A, null] is, ints
    beyond 2^25, NaN, objects without a polygon ...) and the 240-row table of the end-to-end fixture."""
    from oracle import steps as osteps

    g = load_golden("chain_cases.json")
    Q = lambda n: str(tmp_path / n)  # noqa: E731
    write_csv_text(Q("in.csv"), g["input_csv"])
    for run in g["runs"]:
        P.LAST_IO_PATH.clear()
        res = P.process_csv_replace_and_filter(Q("in.csv"), Q("p.csv"), Q("x.csv"), Q("hi.csv"), Q("lo.csv"), run["min_boxes"], run["thr"],
                                               backend=backend)
        assert P.LAST_IO_PATH["replace_iou"] == "fused-native"
        assert res == {"filtered_rows": g["result"]["filtered_rows"], "excluded_rows": g["result"]["excluded_rows"], "excluded_output": Q("x.csv")}
        assert read_text(Q("p.csv")) == g["processed_csv"] and read_text(Q("x.csv")) == g["excluded_csv"]
        assert read_text(Q("hi.csv")) == run["high_csv"], (run["min_boxes"], run["thr"])
        assert read_text(Q("lo.csv")) == run["other_csv"], (run["min_boxes"], run["thr"])
        # the frame twin: same rows
        df = pd.read_csv(Q("in.csv"), encoding="utf-8-sig")
        stats = {}
        kept, excluded, high, other = P.replace_and_filter_frame(df, run["min_boxes"], run["thr"], backend, stats)
        assert stats["fused_launches"] >= 1 and stats["host_rows"] >= 1            # big_ints_* rows: CPython decides
        want_hi = pd.read_csv(Q("hi.csv"), encoding="utf-8-sig", dtype={"source": str})["source"].tolist()
        assert high["source"].tolist() == want_hi and len(high) + len(other) == len(kept) == g["result"]["filtered_rows"]
        assert kept[P.BBOX_COL].isna().sum() == 2 and len(excluded) == 1
        okept, _, oexc = osteps.replace_frame(df)                      # values AND dtypes of the three new columns
        pd.testing.assert_frame_equal(kept, okept)
        pd.testing.assert_frame_equal(excluded, oexc)
    # the end-to-end fixture (240 rows through the reference): filtered -> processed / excluded / high / other, byte for byte
    write_csv_text(Q("filtered.csv"), golden_csv_text("e2e_filtered.csv.gz"))
    P.LAST_IO_PATH.clear()
    P.process_csv_replace_and_filter(Q("filtered.csv"), Q("processed.csv"), Q("excluded.csv"), Q("high.csv"), Q("other.csv"), 2, 0.98,
                                     backend=backend)
    assert P.LAST_IO_PATH["replace_iou"] == "fused-native"
    for n in ("processed", "excluded", "high", "other"):
        assert read_text(Q(n + ".csv")) == golden_csv_text(f"e2e_{n}.csv.gz"), n
    # random polygon tables with empty polygons planted: fused twin == CPU port of the two reference steps in sequence
    rng = np.random.default_rng(5)
    cells = []
    for _ in range(300):
        objs = []
        for _b in range(int(rng.integers(0, 7))):
            kind = rng.random()
            if kind < 0.15:
                objs.append({"polygon": {"ptList": []}})
            elif kind < 0.2:
                objs.append({"name": "no polygon"})
            else:
                x, y = (int(v) for v in rng.integers(0, 3, size=2) * 200)
                objs.append({"polygon": {"ptList": [{"x": x, "y": y}, {"x": x + 100, "y": y}, {"x": x + 100, "y": y + 100 - int(rng.integers(0, 4))}]}})
        cells.append(json.dumps({"objects": objs}))
    df = pd.DataFrame({"source": [f"s{i}" for i in range(len(cells))], P.ANNOTATION_COL: cells})
    for mb, thr in ((2, 0.98), (3, 0.5), (0, 0.0)):
        _, _, high, other = P.replace_and_filter_frame(df, mb, thr, backend)
        okept, oproj, _ = osteps.replace_frame(df)
        ohi, olo = osteps.iou_filter_frame(oproj, mb, thr)
        assert high["source"].tolist() == ohi["source"].tolist() and other["source"].tolist() == olo["source"].tolist(), (mb, thr)
        assert high[P.BBOX_COL].tolist() == ohi[osteps.NEW_COL].tolist()
    # width / height columns: all ints -> int64, ints and gaps -> float64, nothing at all -> object, a string among them -> object
    for ws in ([640, 480, 1], [640, None, 1.5], [None, None, None], [640, "640", None], [2 ** 60, 1, 2]):
        cells = [json.dumps({"objects": [], **({"width": w, "height": w} if w is not None else {})}) for w in ws]
        df = pd.DataFrame({"source": ["a", "b", "c"], P.ANNOTATION_COL: cells})
        kept, _, _, _ = P.replace_and_filter_frame(df, 2, 0.98, backend)
        okept, _, _ = osteps.replace_frame(df)
        pd.testing.assert_frame_equal(kept, okept)


# ---------------------------------------------------------------------------------- CPU runs
def test_chain_golden(oracle_backend, tmp_path):
    run_chain_golden(oracle_backend, tmp_path)


def test_replace_golden(oracle_backend, tmp_path):
    run_replace_golden(oracle_backend, tmp_path)


def test_iou_golden(oracle_backend, tmp_path):
    run_iou_golden(oracle_backend, tmp_path)


def test_dedup_golden(oracle_backend, tmp_path):
    run_dedup_golden(oracle_backend, tmp_path)


def test_ref_filter_golden(oracle_backend, tmp_path):
    run_ref_filter_golden(oracle_backend, tmp_path)


def test_split_golden(oracle_backend):
    run_split_golden(oracle_backend)


def test_e2e_golden(oracle_backend, tmp_path):
    run_e2e_golden(oracle_backend, tmp_path)


def test_error_conventions(oracle_backend, tmp_path):
    """reference processor.py:118-137, :174-192, :237-247, :380-387, :668-671"""
    be = oracle_backend
    with pytest.raises(FileNotFoundError):
        P.deduplicate_csv_by_source(str(tmp_path / "missing.csv"), backend=be)
    txt = tmp_path / "a.txt"
    txt.write_text("source\n1\n")
    with pytest.raises(ValueError):
        P.deduplicate_csv_by_source(str(txt), backend=be)
    nos = tmp_path / "nosource.csv"
    nos.write_text("a,b\n1,2\n")
    with pytest.raises(KeyError):
        P.deduplicate_csv_by_source(str(nos), None, backend=be)
    with pytest.raises(ValueError):
        P.dedup_keep_mask(pd.Series(["a"]), keep="bogus", backend=be)
    with pytest.raises(FileNotFoundError):
        P.remove_duplicates_between_csv(str(nos), str(tmp_path / "missing.csv"), backend=be)
    with pytest.raises(KeyError):
        P.remove_duplicates_between_csv(str(nos), str(nos), str(tmp_path / "o.csv"), backend=be)
    assert P.process_csv_replace_ptlist(str(tmp_path / "missing.csv"), backend=be) is None
    assert P.process_csv_replace_ptlist(str(nos), str(tmp_path / "o.csv"), backend=be) is None
    assert P.filter_by_box_count_and_iou(str(tmp_path / "missing.csv"), backend=be) is None
    assert P.filter_by_box_count_and_iou(str(nos), str(tmp_path / "h.csv"), str(tmp_path / "o.csv"), backend=be) is None
    assert not (tmp_path / "h.csv").exists()
    with pytest.raises(FileNotFoundError):
        P.split_dataset_by_rules(str(tmp_path / "missing.csv"), str(nos), str(tmp_path / "out"), backend=be)


def test_no_device_means_failure_not_fallback(monkeypatch, tmp_path):
    """Without the HIP library / a gfx950 device the product raises; it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu suite")
    from deal_yolo_daya_amd import _native
    with pytest.raises(_native.NativeUnavailable):
        P.dedup_keep_mask(pd.Series(["a", "b", "a"]))
    with pytest.raises(_native.NativeUnavailable):
        P.replace_ptlist_cells(['{"objects": []}'])


def test_flatten_layout():
    cells = ['{"objects": [{"polygon": {"ptList": [{"x": 1, "y": 2}, {"x": 3.5, "y": -4}]}}, {"polygon": {}}]}',
             None, '{"objects": [', '{"objects": [{"polygon": {"ptList": [{"x": true, "y": 9007199254740993}]}}]}']
    b = flatten.flatten_polygons(cells)
    assert b.xy.tolist() == [[1.0, 2.0], [3.5, -4.0]]
    assert b.pt_off.tolist() == [0, 2, 2, 2]
    assert b.cell_box_off.tolist() == [0, 2, 2, 2, 3]
    assert b.stats["host_boxes"] == 1 and list(b.host_boxes) == [2]
    assert b.docs[1] is None and b.docs[2] is None


def test_synth_is_regular_and_deterministic(oracle_backend):
    from deal_yolo_daya_amd import synth
    a, b = synth.generate(300, seed=5), synth.generate(300, seed=5)
    assert np.array_equal(a.xy, b.xy) and np.array_equal(a.pt_off, b.pt_off)
    df = synth.to_frame(a)
    stats = {}
    kept, _ = P.replace_ptlist_frame(df, oracle_backend, stats)
    assert stats["host_boxes"] == 0 and stats["boxes"] == a.n_boxes and stats["points"] == a.n_points
    batch = flatten.flatten_polygons(df[P.ANNOTATION_COL].tolist())
    assert np.array_equal(batch.xy, a.xy) and np.array_equal(batch.pt_off, a.pt_off)   # frame == SoA view
    stats = {}
    mask = P.iou_high_mask(kept[P.BBOX_COL].tolist(), 2, 0.98, oracle_backend, stats)
    assert stats["host_rows"] == 0
    assert 0 < mask.sum() < len(mask)


def test_printed_lines_match_the_reference(oracle_backend, tmp_path, capsys):
    """run_step captures stdout for the UI (reference ui/pages/processing.py): the log lines are part of the drop-in"""
    g = load_golden("e2e_prints.json")
    Q = lambda n: str(tmp_path / n)  # noqa: E731
    for n in ("merged", "ref"):
        write_csv_text(Q(n + ".csv"), golden_csv_text(f"e2e_{n}.csv.gz"))

    def printed(fn, *a, **k):
        capsys.readouterr()
        fn(*a, **k, backend=oracle_backend)
        return capsys.readouterr().out.replace(str(tmp_path), "<TMP>")

    for native_csv in ("1", "0"):
        os.environ["DYD_NATIVE_CSV"] = native_csv
        try:
            assert printed(P.deduplicate_csv_by_source, Q("merged.csv"), Q("dedup.csv")) == g["dedup"]
            assert printed(P.remove_duplicates_between_csv, Q("dedup.csv"), Q("ref.csv"), Q("filtered.csv")) == g["ref_filter"]
            assert printed(P.process_csv_replace_ptlist, Q("filtered.csv"), Q("processed.csv"), Q("excluded.csv")) == g["replace"]
            assert printed(P.filter_by_box_count_and_iou, Q("processed.csv"), Q("high.csv"), Q("other.csv"), 2, 0.98) == g["iou"]
            assert printed(P.process_csv_replace_ptlist, Q("nope.csv"), Q("x.csv"), Q("y.csv")) == g["replace_missing_file"]
            assert printed(P.filter_by_box_count_and_iou, Q("ref.csv"), Q("h2.csv"), Q("o2.csv")) == g["iou_missing_column"]
            assert printed(P.process_csv_replace_ptlist, Q("ref.csv"), Q("x.csv"), Q("y.csv")) == g["replace_missing_column"]
        finally:
            os.environ.pop("DYD_NATIVE_CSV", None)


def test_exception_messages_match_the_reference(oracle_backend, tmp_path):
    g = load_golden("e2e_prints.json")["errors"]
    Q = lambda n: str(tmp_path / n)  # noqa: E731
    write_csv_text(Q("merged.csv"), golden_csv_text("e2e_merged.csv.gz"))
    (tmp_path / "a.txt").write_text("source\n1\n")
    (tmp_path / "nosource.csv").write_text("a,b\n1,2\n")
    calls = {
        "dedup_missing_file": (P.deduplicate_csv_by_source, (Q("missing.csv"),)),
        "dedup_not_csv": (P.deduplicate_csv_by_source, (Q("a.txt"),)),
        "dedup_no_source_column": (P.deduplicate_csv_by_source, (Q("nosource.csv"), None)),
        "ref_missing_ref_file": (P.remove_duplicates_between_csv, (Q("nosource.csv"), Q("missing.csv"))),
        "ref_missing_main_file": (P.remove_duplicates_between_csv, (Q("missing.csv"), Q("nosource.csv"))),
        "ref_not_csv": (P.remove_duplicates_between_csv, (Q("a.txt"), Q("nosource.csv"))),
        "ref_no_column": (P.remove_duplicates_between_csv, (Q("nosource.csv"), Q("nosource.csv"), Q("o.csv"))),
        "ref_no_column_in_ref": (P.remove_duplicates_between_csv, (Q("merged.csv"), Q("nosource.csv"), Q("o.csv"))),
        "split_missing_input": (P.split_dataset_by_rules, (Q("missing.csv"), Q("nosource.csv"), Q("out"))),
        "split_missing_rules": (P.split_dataset_by_rules, (Q("merged.csv"), Q("missing.xlsx"), Q("out"))),
    }
    assert set(calls) == set(g)
    for native_csv in ("1", "0"):
        os.environ["DYD_NATIVE_CSV"] = native_csv
        try:
            for name, (fn, args) in calls.items():
                with pytest.raises(Exception) as info:
                    fn(*args, backend=oracle_backend)
                assert [type(info.value).__name__, str(info.value).replace(str(tmp_path), "<TMP>")] == g[name], (name, native_csv)
        finally:
            os.environ.pop("DYD_NATIVE_CSV", None)


def test_key_columns_of_any_hashable_match_pandas(oracle_backend):
    """drop_duplicates accepts any hashable cell (reference processor.py:140): str, numbers that compare equal across types,
    tuples, Decimal, None / NaN — the keep-mask must be pandas' own for every keep mode; an unhashable cell raises TypeError"""
    from decimal import Decimal
    cols = [
        ["a", "b", "a", np.nan, "b", np.nan, "c"],             # read_csv marks every missing cell with NaN
        ["1.0", 1.0, 1, True, "1", 2, 2.0, "n1.0"],
        [(1, 2), (1, 2), "x", (2, 1), Decimal(1), 1, 1.0, None, frozenset([1]), frozenset([1])],
        [2 ** 60, 2 ** 60 + 1, float(2 ** 60), "s", "s"],
    ]
    for vals in cols:
        col = pd.Series(vals, dtype=object, name="source")
        for keep in ("first", "last", False):
            want = ~col.duplicated(keep=keep).to_numpy()
            assert np.array_equal(P.dedup_keep_mask(col, keep, oracle_backend, verify=False), want), (vals, keep)
            assert np.array_equal(P.dedup_keep_mask(col, keep, oracle_backend), want), (vals, keep)       # verify is the default
    with pytest.raises(TypeError):
        P.dedup_keep_mask(pd.Series(["a", [1, 2]], dtype=object, name="source"), "first", oracle_backend)


def test_verify_mode_catches_a_hash_collision(oracle_backend):
    """a backend whose hash maps two different URLs to one key: without verify the second row is dropped, with verify (the
    default) the collision is detected — the bytes of a row and of the row it was matched to differ — and the exact masks come back"""
    class Colliding:
        def __getattr__(self, name):
            return getattr(oracle_backend, name)

        def hash128(self, data, off):
            h = oracle_backend.hash128(data, off)
            h[1] = h[0]                                  # rows 0 and 1 collide
            return h

    main = pd.Series(["http://a/1.jpg", "http://a/2.jpg", "http://a/3.jpg", "http://a/1.jpg"], name="source")
    be = Colliding()
    assert P.dedup_keep_mask(main, "first", be, verify=False).tolist() == [True, False, True, False]   # the collision swallowed row 1
    P.VERIFY_EVENTS.clear()
    assert P.dedup_keep_mask(main, "first", be).tolist() == [True, True, True, False]
    assert P.dedup_keep_mask(main, False, be).tolist() == [False, True, True, False]
    assert len(P.VERIFY_EVENTS) == 2 and P.VERIFY_EVENTS[0][0] == "dedup"
    assert P.dedup_frame(pd.DataFrame({"source": main}), "first", be)["source"].tolist() == main.tolist()[:3]
    P.VERIFY_EVENTS.clear()
    clean = pd.Series([f"http://a/{k % 700}.jpg" for k in range(3000)] + [None, None], name="source")
    assert np.array_equal(P.dedup_keep_mask(clean, "last", oracle_backend), ~clean.duplicated(keep="last").to_numpy()) and not P.VERIFY_EVENTS
    ref = pd.Series(["http://a/1.jpg"], name="source")

    class CollidingRef(Colliding):
        def hash128(self, data, off):
            h = oracle_backend.hash128(data, off)
            if len(h) == 4:
                h[1] = h[0]
            return h

    assert P.ref_hit_mask(main, ref, CollidingRef(), verify=False).tolist() == [True, True, False, True]
    assert P.ref_hit_mask(main, ref, CollidingRef()).tolist() == [True, False, False, True]
    assert P.VERIFY_EVENTS and P.VERIFY_EVENTS[-1][0] == "ref_filter"
    P.VERIFY_EVENTS.clear()
    ref2 = pd.Series(["http://a/5.jpg", None, "nan", "None", "http://a/699.jpg"])
    want = clean.astype(str).isin(set(ref2.dropna().astype(str))).to_numpy()
    assert np.array_equal(P.ref_hit_mask(clean, ref2, oracle_backend), want) and want.sum() == 11 and not P.VERIFY_EVENTS


def test_the_iou_step_takes_the_table_the_replace_step_parked(oracle_backend, tmp_path, monkeypatch):
    """the processing page's two buttons (reference ui/pages/processing.py:580-598): replace, then IoU filter on the file it wrote.
    The second step writes its CSVs from the parked table — the same bytes as the route that reads the file back — and only
    when file and thresholds are the ones the table was parked for."""
    from deal_yolo_daya_amd import synth
    Q = lambda n: str(tmp_path / n)  # noqa: E731
    df = synth.to_frame(synth.generate(1500, seed=41))
    df.loc[[3, 77], P.ANNOTATION_COL] = None
    df.to_csv(Q("in.csv"), index=False, encoding="utf-8-sig")
    P.clear_step_cache()
    P._STEP_CACHE["params"] = (2, 0.98)

    def two_steps(tag, mb=2, thr=0.98, touch=False):
        res = P.process_csv_replace_ptlist(Q("in.csv"), Q(f"p_{tag}.csv"), Q(f"e_{tag}.csv"), backend=oracle_backend)
        if touch:
            with open(Q(f"p_{tag}.csv"), "ab") as f:
                f.write(b"")
            os.utime(Q(f"p_{tag}.csv"), ns=(1, 1))
        P.filter_by_box_count_and_iou(Q(f"p_{tag}.csv"), Q(f"h_{tag}.csv"), Q(f"o_{tag}.csv"), mb, thr, backend=oracle_backend)
        return res, P.LAST_IO_PATH["iou"]

    res_c, how_c = two_steps("c")
    assert how_c == "cached" and res_c["filtered_rows"] == 1498 and res_c["excluded_rows"] == 2
    monkeypatch.setenv("DYD_STEP_CACHE_MB", "0")
    res_n, how_n = two_steps("n")
    monkeypatch.delenv("DYD_STEP_CACHE_MB")
    assert how_n == "native" and res_n == {**res_c, "excluded_output": Q("e_n.csv")}
    for kind in "pehos"[:4]:
        assert read_text(Q(f"{kind}_c.csv")) == read_text(Q(f"{kind}_n.csv")), kind
    high = pd.read_csv(Q("h_c.csv"), encoding="utf-8-sig")
    assert 10 < len(high) < 200
    # a file touched after it was written is read back; other thresholds are computed, and remembered for the next replace step
    assert two_steps("t", touch=True)[1] == "native" and read_text(Q("h_t.csv")) == read_text(Q("h_n.csv"))
    assert two_steps("q", 3, 0.5)[1] == "native"
    assert two_steps("r", 3, 0.5)[1] == "cached" and read_text(Q("h_r.csv")) == read_text(Q("h_q.csv")) and read_text(Q("o_r.csv")) == read_text(Q("o_q.csv"))
    # another file of the same content is not the parked one
    P.process_csv_replace_ptlist(Q("in.csv"), Q("p_x.csv"), Q("e_x.csv"), backend=oracle_backend)
    P.filter_by_box_count_and_iou(Q("p_r.csv"), Q("h_y.csv"), Q("o_y.csv"), 3, 0.5, backend=oracle_backend)
    assert P.LAST_IO_PATH["iou"] == "native"
    P.clear_step_cache()
    P._STEP_CACHE["params"] = (2, 0.98)


def test_row_subsets_of_large_frames_keep_dtypes_labels_and_values(oracle_backend):
    """dedup_frame / ref_filter_frame build their result with one threaded take per column once the table is large: the frame
    must be what ``df[mask].reset_index(drop=True)`` / ``df[~hit].copy()`` are (processor.py:144, :199) for every column kind"""
    from deal_yolo_daya_amd import pycells
    rng = np.random.default_rng(0)
    n = 6000
    df = pd.DataFrame({"source": [f"u{i % 3500}" for i in range(n)], "w": rng.integers(0, 9, n), "f": rng.random(n),
                       "cat": pd.Categorical(rng.integers(0, 4, n)), "b": rng.random(n) < 0.5,
                       "s": pd.array([None if i % 7 == 0 else f"x{i}" for i in range(n)], dtype="string"),
                       "o": [None if i % 5 == 0 else (i, "t") for i in range(n)],
                       "d": pd.date_range("2020-01-01", periods=n, freq="h"), "I": pd.array(rng.integers(0, 5, n), dtype="Int64")})
    df.index = pd.Index(rng.permutation(n) * 3, name="lab")
    ref = pd.DataFrame({"source": [f"u{i}" for i in range(0, 3500, 3)] + [None]})
    pycells.set_min_threaded(64)
    try:
        for keep in ("first", "last", False):
            got = P.dedup_frame(df, keep, backend=oracle_backend)
            exp = df.drop_duplicates(subset=["source"], keep=keep).reset_index(drop=True)
            pd.testing.assert_frame_equal(got, exp)
            assert got.equals(exp) and type(got.index) is type(exp.index)
        got = P.ref_filter_frame(df, ref, backend=oracle_backend)
        exp = df[~df["source"].astype(str).isin(set(ref["source"].dropna().astype(str)))].copy()
        pd.testing.assert_frame_equal(got, exp)
        assert got.index.name == "lab" and 0 < len(got) < len(df)
        got.iloc[0, 1] = 777                                       # the result owns its data
        assert df.loc[got.index[0], "w"] != 777 or exp.iloc[0, 1] == 777
    finally:
        pycells.set_min_threaded()
