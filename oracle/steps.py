"""CPU restatement of the reference's five hot-path steps.  TEST INFRASTRUCTURE ONLY.

Single-threaded CPython + pandas, deliberately keeping the reference's cost structure
(``read_csv`` -> ``json.loads`` -> builtin ``min``/``max`` -> ``iterrows`` double loop ->
``to_csv``) so that timing it on the GPU box's host stands in for the reference
(bench.py ``cpu_baseline.kind == "port"``).  Each function names the reference lines it
restates (src/deal_yolo_data/core/processor.py unless noted).  Pinned by tests/golden/.
"""
from __future__ import annotations

import copy
import json
import os
import re
from pathlib import Path

import pandas as pd

ANN_COL = "结果字段-目标检测标签配置"           # processor.py:244
NEW_COL = "新_" + ANN_COL                      # processor.py:283, 384
_LABEL_SEP = re.compile(r"[,，;；|]")          # utils.py:642, 662


# --------------------------------------------------------------------------- a1 :111-164
def dedup_frame(df: pd.DataFrame, keep="first") -> pd.DataFrame:
    return df.drop_duplicates(subset=["source"], keep=keep, ignore_index=True)


def dedup_csv(csv_path, output_file="deduplicate_result.csv", encoding="utf-8-sig", keep="first"):
    if not os.path.exists(csv_path):
        raise FileNotFoundError(csv_path)
    if not csv_path.endswith(".csv"):
        raise ValueError(csv_path)
    try:
        df = pd.read_csv(csv_path, encoding=encoding, parse_dates=False)
    except Exception as e:  # :133-134
        raise Exception(str(e)) from e
    if "source" not in df.columns:
        raise KeyError("source")
    out = dedup_frame(df, keep)
    if output_file is not None:
        d = os.path.dirname(output_file)
        if d:
            os.makedirs(d, exist_ok=True)
        out.to_csv(output_file, index=False, encoding=encoding)
    return out


# --------------------------------------------------------------------------- a2 :166-219
def ref_filter_frame(df_main: pd.DataFrame, df_ref: pd.DataFrame, col="source") -> pd.DataFrame:
    wanted = set(df_ref[col].dropna().astype(str))          # :194
    hit = df_main[col].astype(str).isin(wanted)              # :198
    return df_main[~hit].copy()                              # :199


def ref_filter_csv(main_csv, ref_csv, output_csv="filtered_main.csv", compare_col="source",
                   encoding="utf-8-sig"):
    for p in (main_csv, ref_csv):
        if not os.path.exists(p):
            raise FileNotFoundError(p)
        if not p.endswith(".csv"):
            raise ValueError(p)
    df_main = pd.read_csv(main_csv, encoding=encoding, parse_dates=False)
    df_ref = pd.read_csv(ref_csv, encoding=encoding, parse_dates=False)
    for d in (df_main, df_ref):
        if compare_col not in d.columns:
            raise KeyError(compare_col)
    out = ref_filter_frame(df_main, df_ref, compare_col)
    d = os.path.dirname(output_csv)
    if d:
        os.makedirs(d, exist_ok=True)
    out.to_csv(output_csv, index=False, encoding=encoding)
    return out


# --------------------------------------------------------------------------- a3 :229-319
def bbox_of_ptlist(ptlist):
    """:252-260 — builtin min/max keep the FIRST extremal element and its Python type."""
    pts = [p for p in ptlist if isinstance(p, dict) and "x" in p and "y" in p]
    if not pts:
        return [{"x": None, "y": None}, {"x": None, "y": None}]
    lo_x = min(p["x"] for p in pts)
    hi_x = max(p["x"] for p in pts)
    lo_y = min(p["y"] for p in pts)
    hi_y = max(p["y"] for p in pts)
    return [{"x": lo_x, "y": lo_y}, {"x": hi_x, "y": hi_y}]


def replace_cell(cell):
    """:262-281 — only JSONDecodeError is swallowed; structural surprises raise."""
    try:
        if pd.isna(cell) or not isinstance(cell, str):
            return None
        doc = json.loads(cell)
        rewritten = []
        for obj in doc.get("objects", []):
            if not isinstance(obj, dict):
                continue
            new_obj = obj.copy()
            new_pts = bbox_of_ptlist(obj.get("polygon", {}).get("ptList", []))
            if "polygon" not in new_obj:
                new_obj["polygon"] = {}
            new_obj["polygon"]["ptList"] = new_pts
            rewritten.append(new_obj)
        doc["objects"] = rewritten
        return json.dumps(doc, ensure_ascii=False)
    except json.JSONDecodeError:
        return None


def width_height_of_cell(cell):
    """:285-292 — second parse of the same cell, bare except."""
    try:
        if pd.isna(cell) or not isinstance(cell, str):
            return None, None
        doc = json.loads(cell)
        return doc.get("width"), doc.get("height")
    except:  # noqa: E722  (the reference's bare except)
        return None, None


def replace_frame(df: pd.DataFrame):
    """:249-309 -> (projected frame with NEW_COL/width/height, excluded rows)."""
    kept = df.dropna(subset=[ANN_COL]).copy()
    excluded = df[df[ANN_COL].isna()].copy()
    kept[NEW_COL] = kept[ANN_COL].apply(replace_cell)
    wh = kept[ANN_COL].apply(lambda c: dict(zip(("width", "height"), width_height_of_cell(c))))
    kept["width"] = [d["width"] for d in wh]
    kept["height"] = [d["height"] for d in wh]
    cols = [c for c in ("source", ANN_COL, NEW_COL, "width", "height") if c in kept.columns]
    return kept, kept[cols], excluded


def replace_csv(input_csv_path, output_csv_path="processed_replaced_ptlist.csv",
                excluded_output_file="processed_excluded.csv"):
    try:
        df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
    except Exception:
        return None
    if ANN_COL not in df.columns:
        return None
    kept, projected, excluded = replace_frame(df)
    Path(output_csv_path).parent.mkdir(parents=True, exist_ok=True)
    projected.to_csv(output_csv_path, index=False, encoding="utf-8-sig")
    if excluded_output_file is not None:
        Path(excluded_output_file).parent.mkdir(parents=True, exist_ok=True)
        excluded.to_csv(excluded_output_file, index=False, encoding="utf-8-sig")
    return {"filtered_rows": len(kept), "excluded_rows": len(excluded),
            "excluded_output": excluded_output_file}


# --------------------------------------------------------------------------- a4 :321-407
def pair_iou(p, q):
    """:328-339 — CPython numerics: exact ints, IEEE doubles, true division."""
    ix1 = max(p[0], q[0])
    iy1 = max(p[1], q[1])
    ix2 = min(p[2], q[2])
    iy2 = min(p[3], q[3])
    inter = max(0, ix2 - ix1) * max(0, iy2 - iy1)
    if inter == 0:
        return 0.0
    a1 = (p[2] - p[0]) * (p[3] - p[1])
    a2 = (q[2] - q[0]) * (q[3] - q[1])
    uni = a1 + a2 - inter
    return inter / uni if uni != 0 else 0.0


def boxes_of_cell(cell):
    """:341-366 — any exception truncates the row's box list to the prefix collected."""
    found = []
    try:
        if pd.isna(cell) or not isinstance(cell, str):
            return found
        doc = json.loads(cell)
        for obj in doc.get("objects", []):
            if not isinstance(obj, dict):
                continue
            pts = obj.get("polygon", {}).get("ptList", [])
            if len(pts) != 2:
                continue
            a, b = pts
            if not (isinstance(a, dict) and isinstance(b, dict) and "x" in a and "y" in a
                    and "x" in b and "y" in b):
                continue
            found.append((min(a["x"], b["x"]), min(a["y"], b["y"]),
                          max(a["x"], b["x"]), max(a["y"], b["y"])))
    except Exception:
        pass
    return found


def row_is_high(boxes, min_boxes, thr):
    """:368-376"""
    if len(boxes) < min_boxes:
        return False
    for i in range(len(boxes)):
        for j in range(i + 1, len(boxes)):
            if pair_iou(boxes[i], boxes[j]) >= thr:
                return True
    return False


def iou_filter_frame(df: pd.DataFrame, min_boxes=2, thr=0.98):
    """:389-406 -> (high frame, other frame); iterrows + list-of-Series like the reference."""
    hi, lo = [], []
    for _, row in df.iterrows():
        (hi if row_is_high(boxes_of_cell(row[NEW_COL]), min_boxes, thr) else lo).append(row)
    return pd.DataFrame(hi, columns=df.columns), pd.DataFrame(lo, columns=df.columns)


def iou_filter_csv(input_csv_path, high_iou_csv="high_iou_0.98.csv", other_csv="other_data.csv",
                   min_boxes=2, iou_threshold=0.98):
    try:
        df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
    except Exception:
        return
    if NEW_COL not in df.columns:
        return
    hi, lo = iou_filter_frame(df, min_boxes, iou_threshold)
    Path(high_iou_csv).parent.mkdir(parents=True, exist_ok=True)
    Path(other_csv).parent.mkdir(parents=True, exist_ok=True)
    hi.to_csv(high_iou_csv, index=False, encoding="utf-8-sig")
    lo.to_csv(other_csv, index=False, encoding="utf-8-sig")


# --------------------------------------------------------------------------- a5 :654-831
def split_label_cell(cell):
    """utils.py:635-643"""
    if pd.isna(cell):
        return []
    text = str(cell).strip()
    if not text:
        return []
    return [t.strip() for t in _LABEL_SEP.split(text) if t.strip()]


def split_object_labels(name):
    """utils.py:659-662"""
    if not name:
        return []
    return [t.strip() for t in _LABEL_SEP.split(str(name)) if t.strip()]


def parse_objects(cell):
    """utils.py:645-657"""
    if pd.isna(cell) or not isinstance(cell, str) or not cell:
        return None, [], "空数据"
    try:
        doc = json.loads(cell)
        objs = doc.get("objects", [])
        if not isinstance(objs, list):
            return doc, [], "objects不是列表"
        return doc, objs, None
    except json.JSONDecodeError:
        return None, [], "JSON解析失败"
    except Exception as e:
        return None, [], str(e)


def rules_to_map(rules_df: pd.DataFrame, rule_mode="wide", label_col=None, category_col=None):
    """:688-703"""
    mapping = {}
    if rule_mode == "wide":
        for col in rules_df.columns:
            cat = str(col).strip()
            if not cat:
                continue
            for cell in rules_df[col].dropna():
                for lab in split_label_cell(cell):
                    mapping[lab] = cat
    elif rule_mode == "two_column":
        for _, r in rules_df.iterrows():
            lab = str(r.get(label_col, "")).strip()
            cat = str(r.get(category_col, "")).strip()
            if lab and cat and lab.lower() != "nan" and cat.lower() != "nan":
                mapping[lab] = cat
    return mapping


def split_frames(df: pd.DataFrame, label_to_category: dict, json_columns=None, train_ratio=0.8,
                 val_ratio=0.1, test_ratio=0.1, random_seed=42):
    """:673-676, :680-685, :705-818 without the Excel I/O.

    Returns {"categories": {cat: (train, val, test)}, "unclassified": frame,
             "split_counts": frame, "category_counts": {cat: n}}."""
    s = train_ratio + val_ratio + test_ratio
    train_ratio, val_ratio, test_ratio = train_ratio / s, val_ratio / s, test_ratio / s
    if json_columns is None:
        json_columns = [c for c in (NEW_COL, ANN_COL) if c in df.columns]
    per_cat, unclassified, counts = {}, [], []
    for _, row in df.iterrows():
        cell = None
        for c in json_columns:
            if c in row and isinstance(row[c], str) and row[c]:
                cell = row[c]
                break
        doc, objs, err = parse_objects(cell)
        if err or not objs:
            why = err or "标注字段objects为空"
            r = row.copy()
            r["无法分类原因"] = why
            unclassified.append(r)
            counts.append({"source": row.get("source"), "原始标签组合": "", "拆分条数": 0,
                           "是否可分类": "否", "无法分类原因": why})
            continue
        seen = set()
        for o in objs:
            if isinstance(o, dict) and o.get("name"):
                seen.update(split_object_labels(o.get("name")))
        combo = "，".join(sorted(seen)) if seen else ""
        n_out, reasons, hit = 0, set(), False
        for o in objs:
            if not isinstance(o, dict):
                continue
            labs = split_object_labels(o.get("name"))
            if not labs:
                r = row.copy()
                r["无法分类原因"] = "标注框缺少name字段"
                unclassified.append(r)
                continue
            for lab in labs:
                if lab not in label_to_category:
                    r = row.copy()
                    r["无法分类原因"] = f"标签{lab}未在规则中定义"
                    r["无法分类标签"] = lab
                    unclassified.append(r)
                    reasons.add(f"标签{lab}未在规则中定义")
                    continue
                cat = label_to_category[lab]
                r = row.copy()
                one = copy.deepcopy(o)
                one["name"] = lab
                slim = {k: v for k, v in doc.items() if k != "objects"}
                slim["objects"] = [one]
                text = json.dumps(slim, ensure_ascii=False)
                for c in json_columns:
                    if c in df.columns:
                        r[c] = text
                r["分类标签"] = lab
                r["分类类别"] = cat
                r["原始标签组合"] = combo
                per_cat.setdefault(cat, []).append(r)
                hit = True
                n_out += 1
        if not hit:
            r = row.copy()
            r["无法分类原因"] = "；".join(sorted(reasons)) if reasons else "标签无法匹配规则"
            unclassified.append(r)
        status = "否" if not hit else ("部分可分类" if reasons else "是")
        counts.append({"source": row.get("source"), "原始标签组合": combo, "拆分条数": n_out,
                       "是否可分类": status, "无法分类原因": "；".join(sorted(reasons))})
    out, cat_counts = {}, {}
    for cat, rows in per_cat.items():
        if not rows:
            continue
        cat_counts[cat] = len(rows)
        f = pd.DataFrame(rows).sample(frac=1, random_state=random_seed).reset_index(drop=True)
        n = len(f)
        a = int(n * train_ratio)
        b = int(n * val_ratio)
        out[cat] = (f.iloc[:a], f.iloc[a:a + b], f.iloc[a + b:])
    return {"categories": out, "unclassified": pd.DataFrame(unclassified),
            "split_counts": pd.DataFrame(counts), "category_counts": cat_counts}


# ------------------------------------------------- f4  YOLO label lines  utils.py:681-710, processor.py:1046-1052
def extract_boxes_with_labels(json_str):
    """utils.py:681-710: (label, min x, min y, max x, max y) per named object with a non-empty ptList; x and
    y lists are gathered independently; ANY exception ends the scan and keeps the boxes found so far."""
    found = []
    try:
        if pd.isna(json_str) or not isinstance(json_str, str):
            return found
        for obj in json.loads(json_str).get("objects", []):
            if not isinstance(obj, dict):
                continue
            label = obj.get("name")
            if not label:
                continue
            pts = obj.get("polygon", {}).get("ptList", [])
            if not pts:
                continue
            xs = [p.get("x") for p in pts if isinstance(p, dict) and "x" in p]
            ys = [p.get("y") for p in pts if isinstance(p, dict) and "y" in p]
            if not xs or not ys:
                continue
            found.append((label, min(xs), min(ys), max(xs), max(ys)))
    except Exception:  # noqa: BLE001  (the reference uses a bare except)
        pass
    return found


def yolo_label_lines(boxes, class_id, width, height):
    """processor.py:1046-1052: 'cid cx cy w h' (normalised, %.6f) for every box with positive width and height"""
    lines = []
    for _, x1, y1, x2, y2 in boxes:
        x1, x2 = min(x1, x2), max(x1, x2)
        y1, y2 = min(y1, y2), max(y1, y2)
        bw = max(x2 - x1, 0.0)
        bh = max(y2 - y1, 0.0)
        if bw <= 0 or bh <= 0:
            continue
        lines.append(f"{class_id} {(x1 + x2) / 2 / width:.6f} {(y1 + y2) / 2 / height:.6f} {bw / width:.6f} {bh / height:.6f}")
    return lines


def yolo_row_text(json_str, label_value, class_id, width, height):
    """One row of a split sheet -> (label-file text or None, skip reason or None), in the reference's order of
    checks (processor.py:1001-1060) for a row whose source and image are present."""
    boxes = [b for b in extract_boxes_with_labels(json_str) if b[0] == label_value]
    if not boxes:
        return None, "无匹配标签框"
    if not width or not height:
        return None, "缺少图像尺寸"
    lines = yolo_label_lines(boxes, class_id, width, height)
    if not lines:
        return None, "标注框无效"
    return "\n".join(lines), None


# ------------------------------------------------- f3  merge  processor.py:26-109
def merge_folder(folder_path, output_file="merged_csv.csv", encoding="utf-8-sig", chunk_size=100000, progress_callback=None):
    """processor.py:26-109 restated: chunked read_csv of every *.csv (text-mode handle, errors="ignore"), a
    source_file column, to_csv append; a failing file is reported and skipped; -> total rows or None."""
    if not os.path.exists(folder_path):
        raise FileNotFoundError(f"文件夹不存在：{folder_path}")
    files = list(Path(folder_path).glob("*.csv"))
    if not files:
        print(f"警告：文件夹 {folder_path} 中未找到CSV文件")
        return None
    print(f"找到 {len(files)} 个CSV文件，开始合并...")
    output_file = str(output_file)
    Path(output_file).parent.mkdir(parents=True, exist_ok=True)
    wrote_header, total, done_bytes = False, 0, 0
    all_bytes = sum(f.stat().st_size for f in files)
    for k, path in enumerate(files, start=1):
        try:
            size = path.stat().st_size
            if progress_callback:
                progress_callback(k, len(files), path.name, total, 0, 0, size, 0, all_bytes, done_bytes)
            rows_here = 0
            with open(path, "r", encoding=encoding, errors="ignore") as fh:
                for j, part in enumerate(pd.read_csv(fh, parse_dates=False, chunksize=chunk_size), start=1):
                    part["source_file"] = os.path.basename(path)
                    part.to_csv(output_file, index=False, encoding=encoding, mode="a" if wrote_header else "w", header=not wrote_header)
                    wrote_header = True
                    rows_here += len(part)
                    total += len(part)
                    here = fh.tell()
                    if progress_callback:
                        progress_callback(k, len(files), path.name, total, rows_here, j, size, here, all_bytes, done_bytes + here)
            print(f"成功读取：{path.name}（{rows_here}行）")
            done_bytes += size
        except Exception as exc:  # noqa: BLE001
            print(f"读取失败 {path.name}：{str(exc)}")
            continue
    if not wrote_header:
        print("错误：没有可合并的有效CSV数据")
        return None
    print(f"\n合并完成！共 {total} 行数据")
    print(f"输出文件：{os.path.abspath(output_file)}")
    return total


# --------------------------------------------------------------------------- label_replace :516-652 (between a4 and a5)
def mapping_to_label_map(mapping_df: pd.DataFrame, old_col=None, new_col=None) -> dict:
    """:532-545  first two columns unless named; blanks and "nan" spellings on either side drop the pair"""
    if not old_col or not new_col:
        cols = list(mapping_df.columns)
        if len(cols) < 2:
            raise ValueError("标签对照表至少需要两列")
        old_col = old_col or cols[0]
        new_col = new_col or cols[1]
    out = {}
    for _, row in mapping_df.iterrows():
        old = str(row.get(old_col, "")).strip()
        new = str(row.get(new_col, "")).strip()
        if old and old.lower() != "nan" and new and new.lower() != "nan":
            out[old] = new
    return out


def replace_label_tokens(raw_name, label_map):
    """utils.py:664-679  tokens replaced one by one, then de-duplicated, sorted and joined with ','"""
    if not raw_name:
        return raw_name, 0, 0
    tokens = split_object_labels(raw_name)
    swapped, hits = [], 0
    for t in tokens:
        if t in label_map:
            swapped.append(label_map[t])
            hits += 1
        else:
            swapped.append(t)
    return ",".join(sorted(set(swapped))), hits, len(tokens)


def label_replace_frame(df: pd.DataFrame, label_map: dict, json_columns=None):
    """:547-609  -> (frame with rewritten cells, counters, diff rows, unmatched label counts)"""
    df = df.copy()
    if json_columns is None:
        json_columns = [c for c in (NEW_COL, ANN_COL) if c in df.columns]
    n = {"total_objects": 0, "total_labels": 0, "replaced_labels": 0, "replaced_objects": 0, "replaced_rows": 0,
         "invalid_json_rows": 0, "missing_name_objects": 0}
    unmatched, diff_rows = {}, []
    for idx, row in df.iterrows():
        touched = False
        for col in json_columns:
            if col not in df.columns:
                continue
            cell = row.get(col)
            if pd.isna(cell) or not isinstance(cell, str) or not cell:
                continue
            try:
                doc = json.loads(cell)
            except json.JSONDecodeError:
                n["invalid_json_rows"] += 1
                continue
            objs = doc.get("objects")
            if not isinstance(objs, list):
                continue
            changes = []
            for obj in objs:
                if not isinstance(obj, dict):
                    continue
                n["total_objects"] += 1
                raw = obj.get("name")
                if raw is None:
                    n["missing_name_objects"] += 1
                    continue
                for lbl in split_object_labels(raw):
                    if lbl not in label_map:
                        unmatched[lbl] = unmatched.get(lbl, 0) + 1
                new_name, hits, count = replace_label_tokens(raw, label_map)
                n["total_labels"] += count
                if hits > 0:
                    obj["name"] = new_name
                    n["replaced_labels"] += hits
                    n["replaced_objects"] += 1
                    touched = True
                if raw != new_name:
                    changes.append((raw, new_name))
            doc["objects"] = objs
            df.at[idx, col] = json.dumps(doc, ensure_ascii=False)
            if changes:
                diff_rows.append({"source": row.get("source"), "column": col, "before": "；".join([c[0] for c in changes]),
                                  "after": "；".join([c[1] for c in changes])})
        if touched:
            n["replaced_rows"] += 1
    return df, n, diff_rows, unmatched


def label_replace_csv(input_csv_path, mapping_df, output_csv_path, old_col=None, new_col=None, json_columns=None,
                      diff_excel_path=None, unmatched_excel_path=None, sample_size=30):
    """:516-652 with the mapping sheet handed over as a frame (the Excel layer is pandas' own)"""
    df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
    label_map = mapping_to_label_map(mapping_df, old_col, new_col)
    out, n, diff_rows, unmatched = label_replace_frame(df, label_map, json_columns)
    output_csv_path = Path(output_csv_path)
    output_csv_path.parent.mkdir(parents=True, exist_ok=True)
    out.to_csv(output_csv_path, index=False, encoding="utf-8-sig")
    sheets = {}
    if diff_excel_path:
        sheets["diff"] = pd.DataFrame(diff_rows)
    if unmatched_excel_path:
        sheets["unmatched"] = (pd.DataFrame([{"标签": k, "数量": v} for k, v in unmatched.items()]).sort_values("数量", ascending=False)
                               if unmatched else pd.DataFrame(columns=["标签", "数量"]))
    summary = {"total_rows": len(df), "replaced_rows": n["replaced_rows"], "total_objects": n["total_objects"],
               "replaced_objects": n["replaced_objects"], "total_labels": n["total_labels"], "replaced_labels": n["replaced_labels"],
               "invalid_json_rows": n["invalid_json_rows"], "missing_name_objects": n["missing_name_objects"],
               "mapping_size": len(label_map), "unmatched_labels": len(unmatched)}
    return {"output_csv": output_csv_path, "summary": summary, "sheets": sheets, "sample_diff": diff_rows[:sample_size]}


# --------------------------------------------------------------------------- summaries :833-891, :1089-1162
_UNDEFINED_LABEL = re.compile(r"^标签(.+?)(未在规则中定义)$")      # :860


def unclassified_sheets(df: pd.DataFrame) -> dict:
    """:852-885  the three sheets of unclassified_summary.xlsx"""
    df = df.copy()
    reason_col = "无法分类原因"
    if reason_col not in df.columns:
        df[reason_col] = "未知原因"
    reasons = df[reason_col].fillna("未知原因").value_counts().reset_index()
    reasons.columns = ["原因", "数量"]
    per_label, per_pair = {}, {}
    for _, row in df.iterrows():
        reason = row.get(reason_col, "未知原因")
        labels = split_object_labels(row.get("无法分类标签")) if "无法分类标签" in df.columns else []
        if not labels:
            m = _UNDEFINED_LABEL.match(str(reason))
            if m:
                labels = [m.group(1)]
            else:
                per_label["无标签"] = per_label.get("无标签", 0) + 1
                per_pair[("无标签", reason)] = per_pair.get(("无标签", reason), 0) + 1
                continue
        for lbl in labels:
            per_label[lbl] = per_label.get(lbl, 0) + 1
            per_pair[(lbl, reason)] = per_pair.get((lbl, reason), 0) + 1
    label_summary = pd.DataFrame([{"标签": k, "数量": v} for k, v in per_label.items()]).sort_values("数量", ascending=False)
    pair_summary = pd.DataFrame([{"标签": k[0], "原因": k[1], "数量": v} for k, v in per_pair.items()]).sort_values("数量", ascending=False)
    return {"reason_summary": reasons, "label_summary": label_summary, "reason_label": pair_summary}


def yolo_label_counts(dataset_dirs):
    """:1089-1162  per dataset and split: images per label, boxes per label -> (stats, flat frame)"""
    import yaml

    stats, flat = {}, []

    def share(part, whole):
        return f"{(part / whole * 100):.1f}%" if whole else "0.0%"

    for d in dataset_dirs or []:
        if not d:
            continue
        root = Path(d)
        if not root.exists():
            continue
        names = []
        if (root / "data.yaml").exists():
            try:
                names = yaml.safe_load((root / "data.yaml").read_text(encoding="utf-8")).get("names") or []
            except Exception:
                pass
        per_split, images_all, img_all, box_all = {}, 0, {}, {}
        for split in ["train", "val", "test"]:
            img_counts, box_counts, images = {}, {}, 0
            folder = root / "labels" / split
            if folder.exists():
                for txt in folder.glob("*.txt"):
                    images += 1
                    try:
                        lines = txt.read_text(encoding="utf-8", errors="ignore").splitlines()
                    except Exception:
                        continue
                    seen = set()
                    for line in lines:
                        parts = line.strip().split()
                        if not parts:
                            continue
                        try:
                            cid = int(float(parts[0]))
                            name = names[cid] if cid < len(names) else str(cid)
                            seen.add(name)
                            box_counts[name] = box_counts.get(name, 0) + 1
                        except Exception:
                            continue
                    for name in seen:
                        img_counts[name] = img_counts.get(name, 0) + 1
            per_split[split] = {"total_images": images, "label_counts": img_counts, "box_counts": box_counts}
            images_all += images
            for k, v in img_counts.items():
                img_all[k] = img_all.get(k, 0) + v
            for k, v in box_counts.items():
                box_all[k] = box_all.get(k, 0) + v
            for name in set(img_counts) | set(box_counts):
                flat.append({"数据集": root.name, "split": split, "标签": name, "图片数量": img_counts.get(name, 0),
                             "标注框数量": box_counts.get(name, 0), "占比%": share(img_counts.get(name, 0), images), "split总图片数": images})
        per_split["all"] = {"total_images": images_all, "label_counts": img_all, "box_counts": box_all}
        stats[root.name] = per_split
        for name in set(img_all) | set(box_all):
            flat.append({"数据集": root.name, "split": "all", "标签": name, "图片数量": img_all.get(name, 0), "标注框数量": box_all.get(name, 0),
                         "占比%": share(img_all.get(name, 0), images_all), "split总图片数": images_all})
    return stats, pd.DataFrame(flat)
