"""Host side of the YOLO label-line path without a GPU: box extraction, the order of the skip
reasons, routing of irregular rows to the Python arithmetic, against the golden outputs of the reference
(tests/golden/yolo_cases.json).  The device stage is stood in for by the C oracle."""
import json

import numpy as np

from conftest import load_golden
from deal_yolo_daya_amd.core import processor as P
from deal_yolo_daya_amd.core import utils as U
from oracle import steps as osteps


def _same_value(a, b):
    if isinstance(a, float) and isinstance(b, float):
        return (a != a and b != b) or (a == b and np.signbit(a) == np.signbit(b))
    return type(a) is type(b) and a == b


def test_extract_boxes_matches_reference():
    g = load_golden("yolo_cases.json")
    for name, c in g["cases"].items():
        got = U._extract_boxes_with_labels(c["json"])
        assert len(got) == len(c["boxes"]), name
        for a, b in zip(got, c["boxes"]):
            assert all(_same_value(x, y) for x, y in zip(a, b)), (name, a, b)
    for cell in (None, float("nan"), 5, ""):
        assert U._extract_boxes_with_labels(cell) == osteps.extract_boxes_with_labels(cell) == []


def test_label_texts_match_reference(oracle_backend):
    g = load_golden("yolo_cases.json")
    names = list(g["cases"])
    cases = [g["cases"][n] for n in names]
    stats = {}
    texts, reasons = P.yolo_label_texts([c["json"] for c in cases], [c["label"] for c in cases], [c["class_id"] for c in cases],
                                        [c["width"] for c in cases], [c["height"] for c in cases], oracle_backend, stats)
    for n, c, t, why in zip(names, cases, texts, reasons):
        assert t == c["text"], n
        assert (t is None) == (why is not None), n
        assert why == osteps.yolo_row_text(c["json"], c["label"], c["class_id"], c["width"], c["height"])[1], n
    counted = {}
    for why in reasons:
        if why:
            counted[why] = counted.get(why, 0) + 1
    counted[P.REASON_NO_MATCHING_BOX] += 12              # the fixture's filler rows
    assert counted == g["skipped_reasons"]
    assert stats["python_rows"] >= 3 and stats["device_rows"] > 20      # bools / big ints stay on the host


def test_python_lines_equal_port_on_exotic_rows():
    g = load_golden("yolo_cases.json")
    for name in ("huge_values", "tiny_size_divisor", "big_int_coordinates", "int_beyond_2_pow_52", "bool_coordinates", "at_2_pow_43"):
        c = g["cases"][name]
        boxes = U._extract_boxes_with_labels(c["json"])
        assert "\n".join(P._label_lines_python(boxes, c["class_id"], c["width"], c["height"])) == c["text"], name


def test_image_stem():
    assert U._safe_image_stem("http://h/a b/图 1.jpg?x=1", 7) == "1_7"
    assert U._safe_image_stem("http://h/p/cat.01.jpeg", 3) == "cat.01_3"
    assert U._safe_image_stem("", 4) == "img_4" and U._safe_image_stem(None, 5) == "img_5"
    # values observed from the reference's _safe_image_stem (utils.py:712-724)
    assert U._safe_image_stem("http://h/p/?q=1", 2) == "train_2" and U._safe_image_stem("http://h/x.jpg?a=b.c", 1) == "x.jpg_1"
    assert U._safe_image_stem("/tmp/a/case003.jpg", 9) == "case003_9" and U._safe_image_stem(12.5, 3) == "12_3"
