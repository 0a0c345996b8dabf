"""Differential tests of the native flatten / emit (csrc/host_json.cpp) against CPython's json through
the reference restatement (oracle/steps.py): every cell must give the same text / None / exception
type, whether the native scanner handles it or hands it to flatten.py.  Host code: no GPU needed."""
import json
import math
import random

import numpy as np
import pytest

from conftest import load_golden
from deal_yolo_daya_amd import native_json as nj
from deal_yolo_daya_amd.core import processor as P
from oracle import steps as osteps

NUMBERS = ["0", "-0", "1", "-1", "7", "10", "1920", "-35", "123456789", "9007199254740992", "9007199254740993",
           "-9007199254740993", "12345678901234567890", "0.0", "-0.0", "1.0", "1.5", "10.50", "3.14159", "0.1", "0.30000000000000004",
           "1e5", "1E5", "1e+5", "1.5e-7", "1e-5", "0.0001", "0.00001", "1e15", "1e16", "1e22", "1e23", "123456789012345.6",
           "1234567890123456.7", "5e-324", "1.7976931348623157e308", "1e400", "-1e400", "2.5E+3", "100.0", "1e0", "0e0",
           "4.35", "0.000001", "1.0e-6", "33554432", "33554433", "-33554433", "NaN", "Infinity", "-Infinity"]
STRINGS = ['""', '"a"', '"中文"', '"\\u4e2d\\u6587"', '"\\ud83d\\ude00"', '"😀"', '"a\\"b"', '"back\\\\slash"', '"sl\\/ash"',
           '"tab\\tnl\\ncr\\rbs\\bff\\f"', '"\\u0000\\u001f\\u007f"', '"é\\u00e9"', '"x"', '"y"', '"objects"', '"c1,c2"']
LONE = ['"\\ud800"', '"\\udc00x"', '"\\ud83dabc"']
KEYS = ['"objects"', '"polygon"', '"ptList"', '"x"', '"y"', '"width"', '"height"', '"name"', '"id"', '"k"', '"中"',
        '"\\u0078"', '"a b"']


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)

    def ws(self):
        return self.r.choice(["", "", "", " ", "  ", "\n", "\t", " \r\n "])

    def number(self):
        return self.r.choice(NUMBERS)

    def scalar(self):
        r = self.r.random()
        if r < 0.45:
            return self.number()
        if r < 0.7:
            return self.r.choice(STRINGS)
        if r < 0.72:
            return self.r.choice(LONE)
        return self.r.choice(["true", "false", "null"])

    def value(self, depth=0):
        r = self.r.random()
        if depth > 3 or r < 0.55:
            return self.scalar()
        if r < 0.78:
            return self.obj(depth + 1)
        return self.arr(depth + 1)

    def obj(self, depth, keys=None, vals=None):
        n = self.r.randint(0, 4)
        items = []
        for _ in range(n):
            k = self.r.choice(KEYS)
            items.append(f"{self.ws()}{k}{self.ws()}:{self.ws()}{self.value(depth)}{self.ws()}")
        if self.r.random() < 0.03 and items:
            items.append(items[0])                      # duplicate key
        return "{" + ",".join(items) + "}" if items else "{" + self.ws() + "}"

    def arr(self, depth):
        n = self.r.randint(0, 4)
        return "[" + ",".join(f"{self.ws()}{self.value(depth)}{self.ws()}" for _ in range(n)) + "]"

    def point(self):
        r = self.r.random()
        if r < 0.75:
            items = [f'"x"{self.ws()}:{self.ws()}{self.number()}', f'"y": {self.number()}']
        elif r < 0.85:
            items = [f'"x": {self.scalar()}', f'"y": {self.scalar()}']
        elif r < 0.9:
            items = [f'"x": {self.number()}']
        else:
            return self.value(2)
        if self.r.random() < 0.2:
            items.append(f'{self.r.choice(KEYS)}: {self.value(2)}')
        self.r.shuffle(items)
        return "{" + ", ".join(items) + "}"

    def ptlist(self, two=False):
        r = self.r.random()
        if r < 0.85:
            n = 2 if two and self.r.random() < 0.8 else self.r.randint(0, 6)
            return "[" + ("," + self.ws()).join(self.point() for _ in range(n)) + "]"
        return self.value(2)

    def annotation_object(self, two=False):
        items = []
        if self.r.random() < 0.9:
            pg = []
            if self.r.random() < 0.9:
                pg.append(f'"ptList"{self.ws()}:{self.ws()}{self.ptlist(two)}')
            if self.r.random() < 0.3:
                pg.append(f'{self.r.choice(KEYS)}: {self.value(2)}')
            self.r.shuffle(pg)
            poly = "{" + ", ".join(pg) + "}" if self.r.random() < 0.93 else self.value(2)
            items.append(f'"polygon": {poly}')
        if self.r.random() < 0.7:
            items.append(f'"name": {self.r.choice(STRINGS)}')
        if self.r.random() < 0.3:
            items.append(f'{self.r.choice(KEYS)}: {self.value(2)}')
        self.r.shuffle(items)
        return "{" + ("," + self.ws()).join(items) + "}"

    def cell(self, two=False):
        r = self.r.random()
        if r < 0.04:
            return self.value(0)
        items = []
        if self.r.random() < 0.92:
            if self.r.random() < 0.93:
                n = self.r.randint(0, 5)
                objs = [self.annotation_object(two) if self.r.random() < 0.9 else self.value(1) for _ in range(n)]
                items.append(f'"objects"{self.ws()}:{self.ws()}[' + ("," + self.ws()).join(objs) + "]")
            else:
                items.append(f'"objects": {self.value(1)}')
        for key in ("width", "height"):
            if self.r.random() < 0.6:
                items.append(f'"{key}": {self.scalar() if self.r.random() < 0.8 else self.value(1)}')
        if self.r.random() < 0.4:
            items.append(f'{self.r.choice(KEYS)}: {self.value(1)}')
        self.r.shuffle(items)
        text = self.ws() + "{" + ("," + self.ws()).join(items) + "}" + self.ws()
        m = self.r.random()
        if m < 0.05 and text:                           # corrupt the text
            k = self.r.randrange(len(text))
            text = text[:k] + self.r.choice(['"', "}", ",", "x", "\\", "\x01", ""]) + text[k + self.r.randint(0, 2):]
        elif m < 0.07:
            text = text[: self.r.randrange(len(text) + 1)]
        return text


def _same(a, b):
    if isinstance(a, float) and isinstance(b, float):
        return (math.isnan(a) and math.isnan(b)) or (a == b and math.copysign(1, a) == math.copysign(1, b))
    return type(a) is type(b) and a == b


def _ref_replace(cell):
    try:
        return ("ok", osteps.replace_cell(cell), osteps.width_height_of_cell(cell))
    except Exception as e:  # noqa: BLE001
        return ("raise", type(e).__name__, None)


def _our_replace(cell, backend):
    stats = {}
    try:
        t, w, h = P.replace_ptlist_cells([cell], backend, stats)
        return ("ok", t[0], (w[0], h[0])), stats
    except Exception as e:  # noqa: BLE001
        return ("raise", type(e).__name__, None), stats


@pytest.mark.parametrize("seed", range(12))
def test_replace_fuzz_matches_cpython(oracle_backend, seed):
    g = Gen(seed)
    native = 0
    for _ in range(600):
        cell = g.cell()
        want = _ref_replace(cell)
        got, stats = _our_replace(cell, oracle_backend)
        if want[0] == "ok" and got[0] == "ok" and want[1] is not None:
            try:
                want[1].encode("utf-8")
            except UnicodeEncodeError:
                continue                      # lone surrogate survives json.dumps; to_csv would raise later
        assert got[0] == want[0], cell
        assert got[1] == want[1], cell
        if want[0] == "ok":
            assert _same(got[2][0], want[2][0]) and _same(got[2][1], want[2][1]), cell
        native += stats.get("python_cells", 1) == 0
    assert native > 150                       # the native scanner must be doing a good share of the work


@pytest.mark.parametrize("seed", range(12))
def test_iou_fuzz_matches_cpython(oracle_backend, seed):
    g = Gen(1000 + seed)
    native = 0
    for _ in range(600):
        cell = g.cell(two=True)
        for mb, thr in ((2, 0.98), (1, 0.0)):
            try:
                want = ("ok", osteps.row_is_high(osteps.boxes_of_cell(cell), mb, thr))
            except Exception as e:  # noqa: BLE001
                want = ("raise", type(e).__name__)
            stats = {}
            try:
                got = ("ok", bool(P.iou_high_mask([cell], mb, thr, oracle_backend, stats)[0]))
            except Exception as e:  # noqa: BLE001
                got = ("raise", type(e).__name__)
            assert got == want, (cell, mb, thr)
        native += stats.get("python_cells", 1) == 0
    assert native > 150


def test_batch_order_and_mixed_cells(oracle_backend):
    """a batch mixing regular, irregular, undecodable and missing cells keeps every result in place"""
    g = load_golden("replace_cases.json")
    names = list(g["value_cases"])
    cells = [g["value_cases"][n]["in"] for n in names] + [None, float("nan"), 7]
    stats = {}
    texts, w, h = P.replace_ptlist_cells(cells, oracle_backend, stats)
    assert texts[:len(names)] == [g["value_cases"][n]["out"] for n in names]
    assert texts[len(names):] == [None, None, None]
    assert 0 < stats["python_cells"] < len(names)
    scan = nj.scan_polygons(cells)
    by = dict(zip(names, scan.status.tolist()))
    assert by["int_polygon"] == nj.OK and by["undecodable_json"] == nj.UNDECODABLE
    assert by["big_ints_exact"] == nj.IRREGULAR and by["dup_keys_last_wins"] == nj.IRREGULAR
    assert scan.status[-3:].tolist() == [nj.MISSING] * 3


def test_native_batches_are_stitched_in_order(oracle_backend, monkeypatch):
    """small native batches (as used above 2M cells) give the same per-cell results in the same order"""
    g = load_golden("replace_cases.json")
    cells = [c["in"] for c in g["value_cases"].values()] * 3 + [None]
    whole = P.replace_ptlist_cells(cells, oracle_backend)
    kept_new = [t for t in whole[0]]
    mask_whole = P.iou_high_mask(kept_new, 2, 0.5, oracle_backend)
    monkeypatch.setattr(P, "_NATIVE_CHUNK_CELLS", 7)
    stats = {}
    parts = P.replace_ptlist_cells(cells, oracle_backend, stats)
    assert parts[0] == whole[0] and parts[1] == whole[1] and parts[2] == whole[2]
    assert stats["cells"] == len(cells) and stats["python_cells"] > 0
    assert np.array_equal(P.iou_high_mask(kept_new, 2, 0.5, oracle_backend), mask_whole)


def test_python_path_switch(oracle_backend, monkeypatch):
    g = load_golden("replace_cases.json")
    cells = [c["in"] for c in g["value_cases"].values()]
    a = P.replace_ptlist_cells(cells, oracle_backend)
    monkeypatch.setenv("DYD_NATIVE_JSON", "0")
    stats = {}
    b = P.replace_ptlist_cells(cells, oracle_backend, stats)
    assert a[0] == b[0] and stats["python_cells"] == len(cells)


def test_float_repr_matches_python():
    """number tokens are re-printed exactly like repr(float(token)) / str(int(token))"""
    r = random.Random(5)
    toks = list(NUMBERS[:-3])
    for _ in range(3000):
        m = r.choice([r.random(), r.random() * 1e6, r.random() * 1e-6, r.uniform(-1e300, 1e300), r.randint(-10**6, 10**6) / 100,
                      r.random() * 10 ** r.randint(-30, 30)])
        toks.append(repr(m))
        toks.append(f"{m:.3e}")
        toks.append(f"{m:.2f}")
    cells = ['{"objects": [], "v": [' + ", ".join(toks[i:i + 50]) + "]}" for i in range(0, len(toks), 50)]
    scan = nj.scan_polygons(cells)
    assert (scan.status == nj.OK).all()
    out = scan.emit(np.zeros((0, 4), np.int32))
    for cell, got in zip(cells, out):
        assert got == json.dumps(json.loads(cell), ensure_ascii=False)


# ------------------------------------------------------------------------------------------ split expansion
NAMES = ['"c1"', '"c2"', '"c1,c2"', '"c1，c3；zz"', '" c2 | c1 "', '"c1;;c2"', '""', '" "', '","', '"\\u3000c1\\u3000"', '"c1\\u00a0"',
         '"\\tc2\\n"', '"undefined"', '"中文,c1"', '"中文"', '"a\\"b"', '"c1|c1"', 'null', 'false', 'true', '0', '7', '1.5', '[]', '{}',
         '["c1"]', '{"a": 1}', '"\\ud83d\\ude00"', '"x\\u001fy"', '"c1\\u001f"']
LABEL_MAP = {"c1": "catA", "c2": "catA", "c3": "catB", "中文": "catB", 'a"b': "catC", "😀": "catC"}


class SplitGen(Gen):
    STR_NAMES = [n for n in NAMES if n.startswith('"')]
    OTHER_KEYS = [k for k in KEYS if k != '"name"']

    def annotation_object(self, two=False):
        items = []
        if self.r.random() < 0.85:
            items.append('"name": ' + (self.r.choice(self.STR_NAMES) if self.r.random() < 0.93 else self.r.choice(NAMES)))
        if self.r.random() < 0.6:
            items.append(f'"polygon": {{"ptList": {self.ptlist(two)}}}')
        if self.r.random() < 0.3:
            items.append(self.r.choice(self.OTHER_KEYS) + ": " + self.value(2))
        if self.r.random() < 0.02 and items:
            items.append(items[0])                      # duplicate key
        self.r.shuffle(items)
        return "{" + ("," + self.ws()).join(items) + "}"


def _norm(ex):
    return {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in ex.items()}


@pytest.mark.parametrize("seed", range(10))
def test_split_expansion_fuzz_matches_cpython(seed, monkeypatch):
    g = SplitGen(2000 + seed)
    cells = [g.cell() for _ in range(500)] + [None, "", '{"objects": [{"name": "c1"}], "k": 1e5}']
    for i in range(0, len(cells), 37):
        cells[i] = None if i % 2 else ""
    native = P._expand_rows(cells, LABEL_MAP)
    from deal_yolo_daya_amd import native_json
    st = native_json.split_expand(cells, list(LABEL_MAP)).status
    assert (st == native_json.SP_OK).sum() > 90 and (st == native_json.SP_IRREGULAR).sum() > 5
    monkeypatch.setenv("DYD_NATIVE_JSON", "0")
    plain = P._expand_rows(cells, LABEL_MAP)
    a, b = _norm(native), _norm(plain)
    for key in b:
        assert a[key] == b[key], key
    assert len(b["json"]) > 100 and len(b["unc_row"]) > 100


def test_split_expansion_label_rules():
    """separators, Unicode whitespace stripping, repeated labels, order of records and events"""
    cells = ['{"a": 1, "objects": [{"name": "\\u3000c1 ,\\u00a0c2；zz|c1", "id": 1}, 5, {"id": 2}, {"name": "c3"}], "b": [1.0, "x"]}']
    ex = P._expand_rows(cells, LABEL_MAP)
    assert ex["label"].tolist() == ["c1", "c2", "c1", "c3"]
    assert ex["json"][0] == '{"a": 1, "b": [1.0, "x"], "objects": [{"name": "c1", "id": 1}]}'
    assert ex["json"][3] == '{"a": 1, "b": [1.0, "x"], "objects": [{"name": "c3"}]}'
    assert ex["combo_of_row"].tolist() == ["c1，c2，c3，zz"]
    assert list(zip(ex["unc_row"].tolist(), ex["unc_reason"].tolist(), ex["unc_label"].tolist())) == \
        [(0, "标签zz未在规则中定义", "zz"), (0, "标注框缺少name字段", None)]
    assert ex["verdict"].tolist() == ["部分可分类"] and ex["reasons_of_row"].tolist() == ["标签zz未在规则中定义"]


# ------------------------------------------------------------------------------------------ YOLO labelled boxes
class YoloGen(SplitGen):
    STR_NAMES = ['"c1"', '"c1"', '"c2"', '"\\u0063\\u0031"', '"中文"', '"c1,c2"', '""', '"x"']

    def point(self):
        if self.r.random() < 0.9:
            items = [f'"x": {self.number()}', f'"y": {self.number()}']
            self.r.shuffle(items)
            return "{" + ", ".join(items[: self.r.choice([2, 2, 2, 2, 1])]) + "}"
        return super().point()

    def cell(self, two=False):
        if self.r.random() < 0.8:
            objs = [self.annotation_object() for _ in range(self.r.randint(0, 4))]
            return '{"objects": [' + ", ".join(objs) + '], "width": 5}'
        return super().cell(two)


@pytest.mark.parametrize("seed", range(8))
def test_yolo_texts_fuzz_matches_cpython(oracle_backend, seed, monkeypatch):
    g = YoloGen(3000 + seed)
    r = random.Random(seed)
    cells = [g.cell() for _ in range(400)] + [None, float("nan"), ""]
    labels = [r.choice(["c1", "c1", "c1", "c2", "中文", "c1,c2", "undefined", ""]) for _ in cells]
    cids = [r.randrange(0, 120) for _ in cells]
    widths = [r.choice([1920, 640.0, 1, 1e-3, 0, 33, float("nan"), -100]) for _ in cells]
    heights = [r.choice([1080, 480.5, 2, 0, 77, float("inf"), 1e-9]) for _ in cells]
    # a coordinate CPython cannot do arithmetic on (a str, None) raises out of the reference's loop (:1046-1052 sit
    # outside its try): such cells are dropped from the batch here, the raising itself is checked below
    keep = []
    for i, (c, lab, k, w, h) in enumerate(zip(cells, labels, cids, widths, heights)):
        try:
            osteps.yolo_row_text(c, lab, k, w, h)
            keep.append(i)
        except TypeError:
            with pytest.raises(TypeError):
                P.yolo_label_texts([c], [lab], [k], [w], [h], oracle_backend)
    cells, labels, cids, widths, heights = ([v[i] for i in keep] for v in (cells, labels, cids, widths, heights))
    stats = {}
    native = P.yolo_label_texts(cells, labels, cids, widths, heights, oracle_backend, stats)
    assert stats["python_cells"] < len(cells) * 3 // 4 and stats["device_rows"] > 5
    want = [osteps.yolo_row_text(c, lab, k, w, h) for c, lab, k, w, h in zip(cells, labels, cids, widths, heights)]
    assert native == ([t for t, _ in want], [why for _, why in want])
    monkeypatch.setenv("DYD_NATIVE_JSON", "0")
    plain = P.yolo_label_texts(cells, labels, cids, widths, heights, oracle_backend)
    assert native == plain
    assert any(t for t in plain[0]) and {"无匹配标签框", "缺少图像尺寸", "标注框无效"} <= set(plain[1])


def test_yolo_texts_odd_sizes_take_the_python_path(oracle_backend):
    cell = '{"objects": [{"name": "a", "polygon": {"ptList": [{"x": 1, "y": 2}, {"x": 3, "y": 4}]}}]}'
    texts, reasons = P.yolo_label_texts([cell] * 4, ["a"] * 4, [1, 2, 3, 4], [10, None, "", True], [10, 10, 10, 10], oracle_backend)
    assert texts[0] == "1 0.200000 0.300000 0.200000 0.200000" and texts[3] == "4 2.000000 0.300000 2.000000 0.200000"
    assert reasons[1] == reasons[2] == "缺少图像尺寸"


# ------------------------------------------------------------------------------------------ label_replace step
RELABEL_MAP = {"c1": "g1", "c2": "g1", "c3": "中", "中文": "c1", 'a"b': "q,r", "😀": "z"}


def _relabel_one_by_one(cells, fn):
    """every cell alone, so that a cell that ends the step (exception) does not hide the ones after it"""
    out = []
    for c in cells:
        tot = P._RelabelTotals()
        try:
            res = fn([c], RELABEL_MAP, tot)
            out.append((res, {k: getattr(tot, k) for k in tot.__slots__ if k != "unmatched"}, list(tot.unmatched.items())))
        except Exception as e:  # noqa: BLE001
            out.append((type(e).__name__, str(e)))
    return out


@pytest.mark.parametrize("seed", range(10))
def test_relabel_fuzz_matches_cpython(seed):
    from deal_yolo_daya_amd import native_json
    g = SplitGen(5000 + seed)
    cells = [g.cell() for _ in range(500)] + [None, "", float("nan"), 7, '{"objects": [{"name": "c1;c2 , c1"}, {"name": " ， "}], "k": 1e5}']
    st = native_json.relabel(cells, RELABEL_MAP).status
    assert (st == native_json.RL_REWRITTEN).sum() > 90 and (st == native_json.RL_IRREGULAR).sum() > 5
    native = _relabel_one_by_one(cells, P._relabel_cells_native)
    plain = _relabel_one_by_one(cells, P._relabel_cells_python)
    for i, (a, b) in enumerate(zip(native, plain)):
        assert a == b, (i, cells[i], a, b)
    assert sum(1 for b in plain if not isinstance(b[0], str) and b[0][0][1] is not None) > 20        # cells with a diff
    # and as one batch (counters summed, unmatched labels in first-seen order) over the cells that do not raise
    quiet = [c for c, b in zip(cells, plain) if not isinstance(b[0], str)]
    ta, tb = P._RelabelTotals(), P._RelabelTotals()
    assert P._relabel_cells_native(quiet, RELABEL_MAP, ta) == P._relabel_cells_python(quiet, RELABEL_MAP, tb)
    assert list(ta.unmatched.items()) == list(tb.unmatched.items()) and len(tb.unmatched) >= 3
    assert all(getattr(ta, k) == getattr(tb, k) for k in ta.__slots__)


def test_relabel_rules():
    """separators, Unicode whitespace, replaced labels merged and sorted, names that only change their spelling"""
    cells = ['{"a": 1, "objects": [{"name": "\\u3000c2 ,\\u00a0c1；zz|c1", "id": 1}, 5, {"id": 2}, {"name": "b,a"}, {"name": "c3"}], "b": [1.0, "x"]}']
    tot = P._RelabelTotals()
    (text, before, after, renamed), = P._relabel_cells_native(cells, RELABEL_MAP, tot)
    assert text == '{"a": 1, "objects": [{"name": "g1,zz", "id": 1}, 5, {"id": 2}, {"name": "b,a"}, {"name": "中"}], "b": [1.0, "x"]}'
    assert before == "　c2 , c1；zz|c1；b,a；c3" and after == "g1,zz；a,b；中" and renamed
    assert (tot.total_objects, tot.missing_name_objects, tot.total_labels, tot.replaced_labels, tot.replaced_objects) == (4, 1, 7, 4, 2)
    assert list(tot.unmatched.items()) == [("zz", 1), ("b", 1), ("a", 1)]


def _random_number_token(r):
    """number spellings around every rule of the single-parse lane (csrc/host_json_fast.h): trailing / leading zeros, 15..20
    significant digits, fixed / exponent notation boundaries of float.__repr__, big ints"""
    kind = r.random()
    sign = "-" if r.random() < 0.3 else ""
    digits = lambda n, first="123456789": (r.choice(first) + "".join(r.choice("0123456789") for _ in range(n - 1))) if n else ""  # noqa: E731
    if kind < 0.25:
        return sign + (digits(r.randint(1, 22)) if r.random() < 0.9 else "0")
    if kind < 0.8:
        ip = "0" if r.random() < 0.3 else digits(r.randint(1, 18))
        fr = "0" * r.choice([0, 0, 1, 2, 3, 4, 5, 8]) + "".join(r.choice("0123456789") for _ in range(r.randint(0, 18)))
        fr = (fr or "0") + "0" * r.choice([0, 0, 0, 1, 2, 5])
        return f"{sign}{ip}.{fr}"
    mant = digits(r.randint(1, 17))
    if r.random() < 0.5:
        mant += "." + "".join(r.choice("0123456789") for _ in range(r.randint(1, 6)))
    return f"{sign}{mant}{r.choice('eE')}{r.choice(['', '+', '-'])}{r.randint(0, 30)}"


def test_fast_lane_reprints_numbers_like_cpython(oracle_backend):
    r = random.Random(11)
    cells = []
    # the point spellings around the lean point parser (host_json_fast.h lean_point): json.dumps' own, the compact one, and near
    # misses that must take the general walk (blank before '}', two blanks, y first, a third member, a string coordinate)
    shapes = ['{"x": %s, "y": %s}'] * 6 + ['{"x":%s,"y":%s}', '{"x":%s, "y": %s}', '{"x": %s, "y": %s }', '{"x": %s,  "y": %s}',
                                          '{"y": %s, "x": %s}', '{"x": %s, "y": %s, "z": 1}', '{ "x": %s, "y": %s}']
    for _ in range(4000):
        pts = ", ".join(r.choice(shapes) % (_random_number_token(r), _random_number_token(r)) for _ in range(r.randint(1, 8)))
        cells.append('{"width": %s, "objects": [{"polygon": {"ptList": [%s]}, "score": %s}], "k": [%s, %s]}'
                     % (_random_number_token(r), pts, _random_number_token(r), _random_number_token(r), _random_number_token(r)))
    scan = nj.scan_polygons(cells)
    assert scan.fast_cells > 2000                      # the rest holds ints beyond 2^53 among its coordinates (irregular)
    scan.close()
    for start in range(0, len(cells), 500):            # smaller batches so that a raising cell does not hide the others
        batch = cells[start:start + 500]
        texts, widths, heights = P.replace_ptlist_cells(batch, oracle_backend)
        for cell, got, gw in zip(batch, texts, widths):
            assert got == osteps.replace_cell(cell), cell
            want_w = json.loads(cell)["width"]
            assert _same(gw, want_w) and type(gw) is type(want_w), cell


def test_fast_lane_and_exact_walker_agree(oracle_backend, monkeypatch):
    """the same fuzzed cells through the single-parse lane and with DYD_JSON_FAST=0 (every cell through the exact walker)"""
    cells = [Gen(900 + s).cell() for s in range(3000)]
    for c in list(cells[:300]):
        cells.append(json.dumps(json.loads(c)) if _decodes(c) else c)      # canonically spelled cells: the lane's home ground

    def run():
        scan = nj.scan_polygons(cells)
        box, arg4 = oracle_backend.bbox_minmax(scan.xy, scan.pt_off)
        res = (scan.status.copy(), scan.xy.copy(), scan.pt_off.copy(), scan.cell_box_off.copy(), scan.iou_host.copy(),
               scan.w_kind.copy(), scan.h_kind.copy(), scan.w_val.copy(), scan.h_val.copy(), scan.emit(arg4), scan.fast_cells)
        scan.close()
        return res

    fast = run()
    monkeypatch.setenv("DYD_JSON_FAST", "0")
    exact = run()
    monkeypatch.setenv("DYD_JSON_FAST", "2")           # the lane without its lean point parser: every point by the general walk
    general = run()
    assert fast[-1] > 800 and exact[-1] == 0 and general[-1] == fast[-1]
    for a, b, c in zip(fast[:-2], exact[:-2], general[:-2]):
        assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)
    assert fast[-2] == exact[-2] == general[-2]


def _decodes(c):
    try:
        json.loads(c)
        return True
    except Exception:  # noqa: BLE001
        return False
