"""Label / JSON helpers used by the split step — host-side mirror of the three helpers of the
reference's ``core/utils.py`` that sit on the hot path (same names, same results):

    _split_label_cell     reference utils.py:635-643
    _parse_data_objects   reference utils.py:645-657
    _split_object_labels  reference utils.py:659-662
    safe_filename         reference utils.py:525-529 (names the per-category workbook)
"""
from __future__ import annotations

import json
import re

import pandas as pd

# any of  ,  ，  ;  ；  |   (reference utils.py:642 and :662 use the same class)
_SEPARATORS = re.compile(r"[,，;；|]")


def _tokens(text: str) -> list:
    return [tok for tok in (piece.strip() for piece in _SEPARATORS.split(text)) if tok]


def _split_label_cell(cell_value) -> list:
    """Labels listed in one cell of the rules workbook; NaN / blank cells give []."""
    if pd.isna(cell_value):
        return []
    text = str(cell_value).strip()
    return _tokens(text) if text else []


def _split_object_labels(raw_name) -> list:
    """Labels carried by one object's ``name``; falsy names give []."""
    return _tokens(str(raw_name)) if raw_name else []


def _parse_data_objects(json_str):
    """-> (document, objects, error-text-or-None) with the reference's three error strings."""
    if pd.isna(json_str) or not isinstance(json_str, str) or not json_str:
        return None, [], "空数据"
    try:
        document = json.loads(json_str)
        objects = document.get("objects", [])
    except json.JSONDecodeError:
        return None, [], "JSON解析失败"
    except Exception as exc:  # e.g. a top-level list has no .get
        return None, [], str(exc)
    if not isinstance(objects, list):
        return document, [], "objects不是列表"
    return document, objects, None


def safe_filename(value: str) -> str:
    if not value:
        return "train"
    cleaned = re.sub(r"[^A-Za-z0-9._-]+", "_", value).strip("_")
    return cleaned or "train"
