"""Label / JSON helpers used by the split step — host-side mirror of the three helpers of the
reference's ``core/utils.py`` that sit on the hot path (same names, same results):

    _split_label_cell     reference utils.py:635-643
    _parse_data_objects   reference utils.py:645-657
    _split_object_labels  reference utils.py:659-662
    safe_filename         reference utils.py:525-529 (names the per-category workbook)
    _extract_boxes_with_labels  reference utils.py:681-710 (labelled boxes of a cell, YOLO step)
    _safe_image_stem      reference utils.py:712-724 (label / image file stem)
    _ensure_image_cached  reference utils.py:726-748, download_image utils.py:44-55 (image side of the YOLO step)
"""
from __future__ import annotations

import json
import re
from pathlib import Path

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


def _extract_boxes_with_labels(json_str) -> list:
    """[(name, min x, min y, max x, max y)] of the named objects with a non-empty ptList.

    As in the reference the x and the y values are collected independently (a point may carry only one
    of them), min / max are CPython's first-wins builtins over the decoded values, and the first exception
    of any kind ends the scan, keeping what was collected before it."""
    collected = []
    if not isinstance(json_str, str):          # NaN / None / numbers: nothing to read
        return collected
    try:
        for obj in json.loads(json_str).get("objects", []):
            if not isinstance(obj, dict):
                continue
            name = obj.get("name")
            if not name:
                continue
            points = obj.get("polygon", {}).get("ptList", [])
            if not points:
                continue
            dict_points = [pt for pt in points if isinstance(pt, dict)]
            xs = [pt.get("x") for pt in dict_points if "x" in pt]
            ys = [pt.get("y") for pt in dict_points if "y" in pt]
            if xs and ys:
                collected.append((name, min(xs), min(ys), max(xs), max(ys)))
    except Exception:  # noqa: BLE001 - the reference swallows everything here
        pass
    return collected


def _safe_image_stem(source_url, idx) -> str:
    """'<sanitised file stem>_<row index>' (query string dropped), 'img_<idx>' when there is no source"""
    if not source_url:
        return f"img_{idx}"
    try:
        stem = Path(Path(str(source_url)).name).stem
        stem = stem.split("?")[0] if "?" in stem else stem
        return f"{safe_filename(stem)}_{idx}"
    except Exception:  # noqa: BLE001
        return f"img_{idx}"


def download_image(url: str, save_path: str) -> bool:
    """Fetch one image over HTTP (reference utils.py:44-55): True when the file is there afterwards."""
    import os
    if os.path.exists(save_path):
        return True
    try:
        import requests
        response = requests.get(url, stream=True, timeout=15)
        response.raise_for_status()
        with open(save_path, "wb") as fh:
            fh.write(response.content)
        return True
    except Exception as exc:  # noqa: BLE001
        print(f"\n❌ 图片下载失败 {url}：{exc}")
        return False


def _ensure_image_cached(source_url, cache_dir: Path):
    """Local path -> itself; URL -> cached copy under cache_dir (downloaded once); None on any failure
    (reference utils.py:726-748)."""
    if not source_url:
        return None
    try:
        if Path(source_url).exists():
            return Path(source_url)
        filename = source_url.split("/")[-1]
        if "?" in filename:
            filename = filename.split("?")[0]
        if not filename:
            filename = f"image_{hash(source_url)}.jpg"
        cached = cache_dir / filename
        if cached.exists() and cached.stat().st_size > 0:
            return cached
        download_image(source_url, str(cached))
        if cached.exists():
            return cached
    except Exception:  # noqa: BLE001
        pass
    return None
