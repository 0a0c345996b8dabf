"""Drop-in step functions of the annotation hot path, MI355X-native.

Same names, positional signatures, return values, printed log lines and error behaviour as
the five hot-path functions of the reference's ``src/deal_yolo_data/core/processor.py`` — the
Streamlit page imports them by name (reference ui/pages/processing.py:25-38) and calls them
positionally through ``run_step`` (:548, :566, :584, :598, :630):

    deduplicate_csv_by_source        reference processor.py:111-164   -> K3 + K4
    remove_duplicates_between_csv    reference processor.py:166-219   -> K3 + K5
    process_csv_replace_ptlist       reference processor.py:229-319   -> K1
    filter_by_box_count_and_iou      reference processor.py:321-407   -> K2
    split_dataset_by_rules           reference processor.py:654-831   -> K6 (+ host MT19937)

and, either side of them, ``merge_all_csv_in_folder`` (reference processor.py:26-109, native
CSV hand-off) and the label-line arithmetic of generate_yolo_datasets_from_excels
(reference processor.py:1001-1060) as ``yolo_label_texts`` -> K7.  The rest of what the processing page imports from this
module is here as host-only steps: ``replace_labels_by_mapping`` (pipeline step label_replace, reference :516-652, native
relabeller), ``summarize_unclassified`` (:833-891), ``summarize_yolo_label_counts`` (:1089-1162),
``overwrite_reference_with_result`` (:221-227) and ``download_and_draw_annotations`` (:409-514).

Each step is  flatten (cells -> SoA numpy buffers)  ->  device stage (HIP kernels behind
include/dyd.h)  ->  emit (masks / indices back into pandas).  Every step also has a
DataFrame-level twin (``*_frame``) that skips the CSV hand-off.  The device stage is
mandatory: without libdyd_gfx950.so and a gfx950 GPU the steps raise (``_native``).
"""
from __future__ import annotations

import io
import json
import os
import re
from pathlib import Path
from typing import Optional

import numpy as np
import pandas as pd

from .. import flatten as _fl
from .. import fastcsv as _fc
from .. import native_json as _nj
from .. import pycells as _pycells
from ..backend import resolve as _backend
from .utils import (_ensure_image_cached, _extract_boxes_with_labels, _parse_data_objects, _safe_image_stem,
                    _split_label_cell, _split_object_labels, safe_filename)

ANNOTATION_COL = "结果字段-目标检测标签配置"          # reference processor.py:244
BBOX_COL = "新_" + ANNOTATION_COL                    # reference processor.py:283, :384
_CHUNK_CELLS = 1 << 18                               # cells flattened per device batch (Python path)
LAST_IO_PATH = {}                                    # step -> "native" | "pandas": which CSV path the last call took

# ---- the replace step hands its table to the IoU step ---------------------------------------------------------------------
# The processing page presses two buttons: process_csv_replace_ptlist writes the processed CSV, filter_by_box_count_and_iou reads it
# back (reference ui/pages/processing.py:580-598).  The replace step's native pass already knows everything the IoU step is about to
# recompute — the boxes are its own output (:260 -> :354-362) — so it runs the fused K1+K2 launch with the thresholds the IoU step
# was last called with (the page's defaults, app.py:33-34, until then) and parks the table with its HIGH flags here, keyed by the
# file it wrote: absolute path, size, mtime_ns and a digest of the file's first and last megabyte.  filter_by_box_count_and_iou on
# exactly that file with exactly those thresholds writes its two CSVs from the parked table (LAST_IO_PATH["iou"] == "cached": no
# read, no scan, no launch); anything else — another file, a file touched since, other thresholds, a table above
# DYD_STEP_CACHE_MB (default 16384) — takes the normal route.  One table at most is parked; clear_step_cache() drops it.
import threading as _threading

_STEP_CACHE = {"lock": _threading.Lock(), "entry": None, "params": (2, 0.98)}


def clear_step_cache() -> None:
    with _STEP_CACHE["lock"]:
        entry, _STEP_CACHE["entry"] = _STEP_CACHE["entry"], None
    if entry is not None:
        entry["core"]["scan"].close()


def _file_key(path):
    """(absolute path, size, mtime_ns, digest of the first and last MiB) of a file, or None"""
    import hashlib
    try:
        path = os.path.abspath(str(path))
        st = os.stat(path)
        h = hashlib.blake2b(digest_size=16)
        with open(path, "rb") as f:
            h.update(f.read(1 << 20))
            if st.st_size > (2 << 20):
                f.seek(st.st_size - (1 << 20))
                h.update(f.read(1 << 20))
        return (path, st.st_size, st.st_mtime_ns, h.hexdigest())
    except OSError:
        return None


def _step_cache_limit() -> int:
    try:
        return int(os.environ.get("DYD_STEP_CACHE_MB", "16384")) << 20
    except ValueError:
        return 16384 << 20
VERIFY_EVENTS = []                                   # (step, column, detail): verify=True found a 128-bit hash collision
_NATIVE_CHUNK_CELLS = 1 << 21                        # cells per native scan (2M rows ~ 0.26 G points at 124 pts/row)


# =============================================================================== f3  merge
_READ_CHARS = 262144          # characters the C parser takes from the file handle per read (pandas parsers.pyx)
_HEAVY_BYTES_PER_ROW = 64     # a column averaging more than this per cell is carried natively (never parsed)


class _TellEmulator:
    """What ``f.tell()`` shows after pandas has parsed up to a given row when it reads a text-mode handle in
    blocks of 262144 characters (reference processor.py:80): the byte offset behind the last block read.
    Characters are counted as the text layer delivers them: one per UTF-8 lead byte, a CR LF pair as one."""

    def __init__(self, raw: bytes, bom_len: int, crlf: bool):
        self.b = np.frombuffer(raw, np.uint8)
        self.pos = bom_len
        self.crlf = crlf            # every CR of the file is the first half of a CR LF line end (checked by the tokeniser)

    def _read_block(self):
        need, pos, n = _READ_CHARS, self.pos, len(self.b)
        while need > 0 and pos < n:
            seg = self.b[pos:pos + need]
            chars = int(np.count_nonzero((seg & 0xC0) != 0x80))         # characters that start inside the segment
            if self.crlf:
                chars -= int(np.count_nonzero(seg == 13))               # a CR and its LF arrive as one "\n"
            need -= chars
            pos += len(seg)
        while pos < n and ((self.b[pos] & 0xC0) == 0x80 or (self.crlf and self.b[pos] == 10 and self.b[pos - 1] == 13)):
            pos += 1                                                     # the tail of the last character / its LF
        self.pos = pos

    def after(self, byte_end: int) -> int:
        while self.pos < byte_end and self.pos < len(self.b):
            self._read_block()
        return self.pos


def _merge_file_native(csv_file: Path, output_file: str, encoding: str, chunk_size: int, header_written: bool, on_chunk):
    """One input file of the merge through the native CSV path.  -> rows written, or None when the file is left to
    pandas (nothing has been written then).  on_chunk(rows_in_chunk, chunk_idx, file_bytes) reports progress."""
    if not _fc.enabled() or not _fc._utf8_like(encoding) or chunk_size is None or chunk_size <= 0:
        return None
    raw = csv_file.read_bytes()
    sig = "sig" in encoding.lower()
    bom = len(_fc._BOM) if raw.startswith(_fc._BOM) else 0
    if bom and not sig:
        return None                                      # a BOM read as text becomes part of the first name
    try:
        raw.decode("utf-8")                              # errors="ignore" (:70) drops nothing from valid utf-8
    except UnicodeDecodeError:
        return None
    idx = _fc.CsvIndex.open(np.frombuffer(raw, dtype=np.uint8)[bom:])
    if idx is None:
        return None
    names, n_rows = idx.names, idx.n_rows
    if n_rows == 0 or "source_file" in names:
        return None
    heavy = {}
    for c, nm in enumerate(names):
        if idx.col_bytes(c) >= _HEAVY_BYTES_PER_ROW * n_rows:
            col = idx.extract(c)
            if col is not None:
                heavy[nm] = col
    if not heavy:
        return None                                  # nothing to gain: plain pandas
    light_names = [nm for nm in names if nm not in heavy]
    bounds = [(r0, min(r0 + chunk_size, n_rows)) for r0 in range(0, n_rows, chunk_size)]
    if light_names:
        text = idx.project([names.index(nm) for nm in light_names])
        if text is None:
            return None
        light_iter = pd.read_csv(io.BytesIO(text), encoding="utf-8", usecols=light_names, parse_dates=False,
                                 chunksize=chunk_size)
    else:
        light_iter = (pd.DataFrame(index=pd.RangeIndex(r1 - r0)) for r0, r1 in bounds)
    tell = _TellEmulator(raw, bom, idx.has_cr())
    out_names = names + ["source_file"]
    base = os.path.basename(csv_file)
    written = 0
    for chunk_idx, ((r0, r1), light) in enumerate(zip(bounds, light_iter), start=1):
        if len(light) != r1 - r0:
            raise RuntimeError("native CSV path: chunk sizes disagree")      # cannot happen for an indexed file
        light = light.reset_index(drop=True)
        part = {nm: _fc.Utf8Column(col.data, col.off[r0:r1 + 1], col.na[r0:r1]) for nm, col in heavy.items()}
        columns = [part[nm] if nm in part else light[nm] for nm in names]
        columns.append(pd.Series([base] * (r1 - r0), dtype=object))
        first = not header_written and written == 0
        if not _fc.write_table(output_file, out_names, columns, r1 - r0, encoding=encoding, append=not first, header=first):
            frame = pd.DataFrame({nm: (part[nm].cells(range(r1 - r0)) if nm in part else light[nm]) for nm in names},
                                 columns=names)
            frame["source_file"] = base
            frame.to_csv(output_file, index=False, encoding=encoding, mode="w" if first else "a", header=first)
        written += r1 - r0
        on_chunk(r1 - r0, chunk_idx, tell.after(bom + idx.row_end(r1 - 1)))
    return written


def merge_all_csv_in_folder(
        folder_path,
        output_file="merged_csv.csv",
        encoding="utf-8-sig",
        chunk_size: int = 100000,
        progress_callback=None,
):
    """Drop-in for reference processor.py:26-109: append every *.csv of the folder (chunk by chunk, a
    ``source_file`` column added) to one file; -> total rows, or None when there was nothing to merge.
    Files whose wide columns (the annotation JSON) can be carried as bytes go through the native CSV path —
    those columns are never parsed, the narrow ones are parsed by pandas itself chunk by chunk — anything else
    (other encodings, CR line ends, ragged or quoted-oddly files) takes the reference's pandas loop."""
    if not os.path.exists(folder_path):
        raise FileNotFoundError(f"文件夹不存在：{folder_path}")
    csv_files = list(Path(folder_path).glob("*.csv"))
    if not csv_files:
        print(f"警告：文件夹 {folder_path} 中未找到CSV文件")
        return None
    print(f"找到 {len(csv_files)} 个CSV文件，开始合并...")

    output_file = str(output_file)
    Path(output_file).parent.mkdir(parents=True, exist_ok=True)
    header_written = False
    total_rows = 0
    total_bytes = sum(f.stat().st_size for f in csv_files)
    completed_bytes = 0
    LAST_IO_PATH["merge"] = {}
    if _fc.enabled():
        from .. import _native as _nat
        _nat.load_library()            # a missing / unloadable library is a build problem: raise it here, not as 读取失败 per file

    for file_idx, csv_file in enumerate(csv_files, start=1):
        try:
            file_size = csv_file.stat().st_size
            if progress_callback:
                progress_callback(file_idx, len(csv_files), csv_file.name, total_rows, 0, 0, file_size, 0, total_bytes, completed_bytes)
            state = {"file_rows": 0}

            def on_chunk(rows, chunk_idx, file_bytes):
                nonlocal total_rows, header_written
                header_written = True
                state["file_rows"] += rows
                total_rows += rows
                if progress_callback:
                    progress_callback(file_idx, len(csv_files), csv_file.name, total_rows, state["file_rows"], chunk_idx,
                                      file_size, file_bytes, total_bytes, completed_bytes + file_bytes)

            native_rows = _merge_file_native(csv_file, output_file, encoding, chunk_size, header_written, on_chunk)
            LAST_IO_PATH["merge"][csv_file.name] = "native" if native_rows is not None else "pandas"
            if native_rows is None:
                with open(csv_file, "r", encoding=encoding, errors="ignore") as f:
                    for chunk_idx, df in enumerate(pd.read_csv(f, parse_dates=False, chunksize=chunk_size), start=1):
                        df["source_file"] = os.path.basename(csv_file)
                        df.to_csv(output_file, index=False, encoding=encoding, mode="w" if not header_written else "a",
                                  header=not header_written)
                        on_chunk(len(df), chunk_idx, f.tell())
            print(f"成功读取：{csv_file.name}（{state['file_rows']}行）")
            completed_bytes += file_size
        except Exception as e:  # noqa: BLE001 - the reference reports and moves on to the next file
            print(f"读取失败 {csv_file.name}：{str(e)}")
            continue

    if not header_written:
        print("错误：没有可合并的有效CSV数据")
        return None
    print(f"\n合并完成！共 {total_rows} 行数据")
    print(f"输出文件：{os.path.abspath(output_file)}")
    return total_rows


# =============================================================================== a1  dedup
def dedup_keep_mask(col: pd.Series, keep="first", backend=None, verify: bool = True) -> np.ndarray:
    """Boolean keep-mask of ``drop_duplicates(keep=keep)`` on one key column: K3 hash, K4 mask.

    Equality on the device is equality of 128-bit hashes (MurmurHash3 x64_128 of the cell's canonical bytes); pandas compares
    values (:140-144).  ``verify`` (default) proves that the two agree: the device also names, for every row, the first row with
    the same hash (dyd_dedup_partner), and the host compares the BYTES of each such pair (dyd_host_cells_differ, all cores) —
    equal bytes for every pair means every hash-equal group is a value-equal group, so the mask is pandas' mask.  A pair that
    differs (a collision, accidental or crafted) is recorded in VERIFY_EVENTS and the mask is recomputed by pandas' own value
    comparison.  Costs one more gather per row on the device and one pass over the duplicated rows' bytes on the host."""
    if keep not in ("first", "last", False):
        raise ValueError('keep must be either "first", "last" or False')       # pandas' own message
    be = _backend(backend)
    if len(col) == 0:
        return np.zeros(0, bool)
    data, off, na = _fl.column_key_bytes(col)
    h = be.hash128(data, off)
    if na.any():
        h[na] = _fl.NA_KEY                              # all missing cells are one key (NaN == NaN)
    mask = be.dedup(h, keep).astype(bool)
    if verify:
        if hasattr(be, "dedup_partner"):
            from .. import _native
            partner = be.dedup_partner(h)
            rows = np.flatnonzero(partner != np.arange(len(partner)))
            wrong = 0
            if len(rows):
                mates = partner[rows]
                both_na = na[rows] & na[mates]
                differ = _native.cells_differ(data, off, np.where(both_na, -1, rows), data, off, mates, len(rows)).astype(bool)
                wrong = int((differ | (na[rows] != na[mates])).sum())
        else:                                           # a backend without the partner query: count distinct values instead
            distinct_hashes = int(mask.sum()) if keep in ("first", "last") else int(be.dedup(h, "first").sum())
            wrong = abs(distinct_hashes - int(col.nunique(dropna=False)))
        if wrong:
            VERIFY_EVENTS.append(("dedup", str(col.name), wrong))
            mask = ~col.duplicated(keep=keep).to_numpy()
    return mask


def _frame_rows(df: pd.DataFrame, mask: np.ndarray, keep_labels: bool) -> pd.DataFrame:
    """``df[mask]`` as a new frame (``.reset_index(drop=True)`` unless ``keep_labels``): for a large table one threaded take per
    column (object cells with batched reference counts, csrc/pyhelpers.c) instead of pandas' per-block take."""
    if len(df) >= _pycells.MIN_THREADED and _pycells.available() and df.columns.is_unique and df.columns.nlevels == 1:
        rows = np.flatnonzero(mask)

        def taken(c):
            col = df[c]
            if isinstance(col.dtype, np.dtype) and col.dtype.kind in "Oiufb":
                return _pycells.take(col.to_numpy(), rows, checked=True)
            return col.array.take(rows)                         # extension arrays and datetimes keep their dtype through their own take

        out = pd.DataFrame({c: taken(c) for c in df.columns}, copy=False)
        out.columns = df.columns                                # the same Index object kind / name
        if keep_labels:
            out.index = df.index[rows]
        return out
    return df[mask] if keep_labels else df[mask].reset_index(drop=True)      # (a boolean take is a copy already)


def dedup_frame(df: pd.DataFrame, keep="first", backend=None) -> pd.DataFrame:
    """In-memory twin of the dedup step: rows in original order, index reset (:140-144)."""
    mask = dedup_keep_mask(df["source"], keep, backend)
    return _frame_rows(df, mask, keep_labels=False)


def deduplicate_csv_by_source(
        csv_path: str,
        output_file: Optional[str] = "deduplicate_result.csv",
        encoding: str = "utf-8-sig",
        keep: str = "first",
        verbose: bool = True,
        backend=None,
) -> pd.DataFrame:
    if not os.path.exists(csv_path):
        raise FileNotFoundError(f"CSV文件不存在：{csv_path}")
    if not csv_path.endswith(".csv"):
        raise ValueError(f"文件不是CSV格式：{csv_path}（请传入.csv后缀的文件）")
    table = None
    try:
        table = _fc.read_split(csv_path, [ANNOTATION_COL, BBOX_COL], encoding) if _fc.enabled() else None
        if table is not None and not table.heavy:
            table = None                                   # nothing heavy in this file: plain pandas is as good
        df = table.light if table is not None else pd.read_csv(csv_path, encoding=encoding, parse_dates=False)
    except Exception as e:
        raise Exception(f"读取CSV文件失败：{str(e)}") from e
    LAST_IO_PATH["dedup"] = "native" if table is not None else "pandas"
    if verbose:
        print(f"成功读取CSV文件：{os.path.basename(csv_path)}")
        print(f"读取后原始数据行数：{len(df)}")
    if "source" not in df.columns:
        names = table.names if table is not None else list(df.columns)
        raise KeyError(f"CSV文件中未找到'source'列，请检查列名是否正确（当前列名：{names}）")

    if table is not None:                                  # heavy columns stay flat buffers until the result frame
        rows = np.flatnonzero(dedup_keep_mask(df["source"], keep, backend))
        result = _fc.frame_from_split(table, rows)
    else:
        result = dedup_frame(df, keep, backend)
    if verbose:
        print(f"去重策略：按'source'列保留{keep}条数据")
        print(f"去除重复数据行数：{len(df) - len(result)}")
        print(f"去重后剩余数据行数：{len(result)}")

    if output_file is not None:
        try:
            parent = os.path.dirname(output_file)
            if parent:
                os.makedirs(parent, exist_ok=True)
            columns = ([table.heavy[nm] if nm in table.heavy else table.light[nm] for nm in table.names]
                       if table is not None else None)
            if columns is None or not _fc.write_table(output_file, table.names, columns, table.n_rows, rows=rows,
                                                       encoding=encoding):
                result.to_csv(output_file, index=False, encoding=encoding)
        except Exception as e:
            raise Exception(f"保存去重文件失败：{str(e)}") from e
        if verbose:
            print(f"去重后的文件已保存至：{os.path.abspath(output_file)}")
    return result


# =============================================================================== a2  reference filter
def ref_hit_mask(main_col: pd.Series, ref_col: pd.Series, backend=None, verify: bool = True) -> np.ndarray:
    """``main.astype(str).isin(set(ref.dropna().astype(str)))`` (:194-198): K3 on both, K5.

    ``verify`` (default): a value of the reference set always hits (equal strings hash alike), so only a HIT can be wrong; the
    device names the reference row every hit matched (dyd_isin_partner) and the host compares the two cells' bytes.  Hits whose
    bytes differ are re-checked by value against the reference strings (pandas' isin on those rows) and recorded in VERIFY_EVENTS."""
    be = _backend(backend)
    if len(main_col) == 0:
        return np.zeros(0, bool)
    md, mo = _fl.column_str_bytes(main_col)
    rd, ro = _fl.column_str_bytes(ref_col, drop_na=True)
    hm = be.hash128(md, mo)
    hr = be.hash128(rd, ro) if len(ro) > 1 else np.zeros((0, 2), np.uint64)
    hit = be.isin(hm, hr).astype(bool)
    if verify and hit.any():
        rows = np.flatnonzero(hit)
        if hasattr(be, "isin_partner"):
            from .. import _native
            mates = be.isin_partner(hm, hr)[rows]
            suspect = rows[_native.cells_differ(md, mo, rows, rd, ro, mates, len(rows)).astype(bool) | (mates < 0)]
        else:
            suspect = rows
        if len(suspect):
            true_hit = main_col.iloc[suspect].astype(str).isin(set(ref_col.dropna().astype(str))).to_numpy()
            if hasattr(be, "isin_partner") or not true_hit.all():
                VERIFY_EVENTS.append(("ref_filter", str(main_col.name), int(len(suspect) if hasattr(be, "isin_partner") else (~true_hit).sum())))
            hit[suspect[~true_hit]] = False
    return hit


def ref_filter_frame(df_main: pd.DataFrame, df_ref: pd.DataFrame, compare_col: str = "source",
                     backend=None) -> pd.DataFrame:
    hit = ref_hit_mask(df_main[compare_col], df_ref[compare_col], backend)
    return _frame_rows(df_main, ~hit, keep_labels=True)


def remove_duplicates_between_csv(
        main_csv: str,
        ref_csv: str,
        output_csv: str = "filtered_main.csv",
        compare_col: str = "source",
        encoding: str = "utf-8-sig",
        verbose: bool = True,
        backend=None,
) -> pd.DataFrame:
    for path in (main_csv, ref_csv):
        if not os.path.exists(path):
            raise FileNotFoundError(f"文件不存在：{path}")
        if not path.endswith(".csv"):
            raise ValueError(f"文件不是CSV格式：{path}（请传入.csv后缀文件）")
    table = None
    try:
        table = _fc.read_split(main_csv, [ANNOTATION_COL, BBOX_COL], encoding) if _fc.enabled() else None
        if table is not None and (not table.heavy or compare_col in table.heavy):
            table = None
        df_main = table.light if table is not None else pd.read_csv(main_csv, encoding=encoding, parse_dates=False)
        df_ref = pd.read_csv(ref_csv, encoding=encoding, parse_dates=False)
    except Exception as e:
        raise Exception(f"读取CSV失败：{str(e)}") from e
    LAST_IO_PATH["ref_filter"] = "native" if table is not None else "pandas"
    if verbose:
        print(f"读取主文件：{len(df_main)}行")
        print(f"读取参考文件：{len(df_ref)}行")
    if compare_col not in df_main.columns:
        raise KeyError(f"主文件中未找到列 '{compare_col}'")
    if compare_col not in df_ref.columns:
        raise KeyError(f"参考文件中未找到列 '{compare_col}'")

    if table is not None:
        keep_rows = np.flatnonzero(~ref_hit_mask(df_main[compare_col], df_ref[compare_col], backend))
        kept = _fc.frame_from_split(table, keep_rows)
        kept.index = pd.Index(keep_rows)                   # df_main[~is_dup].copy() keeps the original labels (:199)
    else:
        kept = ref_filter_frame(df_main, df_ref, compare_col, backend)
    if verbose:
        print(f"去重依据列：{compare_col}")
        print(f"参考文件中唯一值数量：{df_ref[compare_col].dropna().astype(str).nunique()}")
        print(f"剔除重复行数：{len(df_main) - len(kept)}")
        print(f"保留行数：{len(kept)}")
    try:
        parent = os.path.dirname(output_csv)
        if parent:
            os.makedirs(parent, exist_ok=True)
        columns = ([table.heavy[nm] if nm in table.heavy else table.light[nm] for nm in table.names]
                   if table is not None else None)
        if columns is None or not _fc.write_table(output_csv, table.names, columns, table.n_rows, rows=keep_rows,
                                                   encoding=encoding):
            kept.to_csv(output_csv, index=False, encoding=encoding)
    except Exception as e:
        raise Exception(f"保存结果失败：{str(e)}") from e
    if verbose:
        print(f"结果已保存至：{os.path.abspath(output_csv)}")
    return kept


# =============================================================================== a3  polygon -> bbox
def _replace_cells_python(cells, be, totals) -> tuple:
    """flatten.py path (CPython json, reference accessor order): used for the cells the native scanner
    calls irregular, and for everything when DYD_NATIVE_JSON=0."""
    texts, widths, heights = [], [], []
    for start in range(0, len(cells), _CHUNK_CELLS):
        batch = _fl.flatten_polygons(cells[start:start + _CHUNK_CELLS])
        if len(batch.pt_off) > 1:
            _, arg4 = be.bbox_minmax(batch.xy, batch.pt_off)
        else:
            arg4 = np.zeros((0, 4), np.int32)
        texts.extend(_fl.emit_polygons(batch, arg4))
        for doc in batch.docs:                         # :285-292 (doc is a dict here, or the step raised)
            widths.append(doc.get("width") if doc is not None else None)
            heights.append(doc.get("height") if doc is not None else None)
        for k in ("boxes", "points", "host_boxes"):
            totals[k] += batch.stats[k]
    return texts, widths, heights


def _replace_cells_native(cells, be, totals):
    """one native scan -> K1 -> native emit pass over `cells` (see replace_ptlist_cells)"""
    try:
        scan = _nj.scan_polygons(cells)
    except UnicodeEncodeError:                         # a lone surrogate somewhere: CPython path for the batch
        totals["python_cells"] += len(cells)
        return _replace_cells_python(cells, be, totals)
    irregular = np.flatnonzero(scan.status == _nj.IRREGULAR)
    totals["python_cells"] += int(len(irregular))
    # irregular cells first: they are the only ones that can raise, and they must raise before any output
    py = _replace_cells_python([cells[i] for i in irregular.tolist()], be, totals) if len(irregular) else ([], [], [])
    if scan.n_boxes:
        _, arg4 = be.bbox_minmax(scan.xy, scan.pt_off)
    else:
        arg4 = np.zeros((0, 4), np.int32)
    texts = scan.emit(arg4)
    widths, heights = scan.width_height(0), scan.width_height(1)
    for col, key in ((widths, "width"), (heights, "height")):   # rare value kinds (str / container / huge int): ask CPython
        for i, v in enumerate(col):
            if v is Ellipsis:
                col[i] = json.loads(cells[i]).get(key)
    for j, i in enumerate(irregular.tolist()):
        texts[i], widths[i], heights[i] = py[0][j], py[1][j], py[2][j]
    totals["boxes"] += scan.n_boxes
    totals["points"] += int(scan.xy.shape[0])
    scan.close()
    return texts, widths, heights


def replace_ptlist_cells(cells, backend=None, stats: Optional[dict] = None) -> tuple:
    """(new JSON text or None, width, height) per annotation cell: flatten -> K1 -> emit.

    Flatten / emit run in the native scanner (csrc/host_json.cpp) for regular cells; the cells it
    classifies as irregular go through flatten.py in row order, so the first exception the reference
    would raise is the one raised here.  Cells are processed in batches of _NATIVE_CHUNK_CELLS so that
    one batch stays far below the 2^31-point limit of the int32 offsets."""
    be = _backend(backend)
    cells = list(cells)
    totals = {"cells": len(cells), "boxes": 0, "points": 0, "host_boxes": 0, "python_cells": 0}
    texts, widths, heights = [], [], []
    if _nj.enabled():
        for start in range(0, len(cells), _NATIVE_CHUNK_CELLS):
            t, w, h = _replace_cells_native(cells[start:start + _NATIVE_CHUNK_CELLS], be, totals)
            texts.extend(t); widths.extend(w); heights.extend(h)
    else:
        totals["python_cells"] = len(cells)
        texts, widths, heights = _replace_cells_python(cells, be, totals)
    if stats is not None:
        stats.update(totals)
    return texts, widths, heights


def replace_ptlist_frame(df: pd.DataFrame, backend=None, stats: Optional[dict] = None):
    """In-memory twin of the replace step -> (kept frame with the three new columns, excluded rows)."""
    kept = df.dropna(subset=[ANNOTATION_COL]).copy()               # :249
    excluded = df[df[ANNOTATION_COL].isna()].copy()                # :250
    texts, widths, heights = replace_ptlist_cells(kept[ANNOTATION_COL].tolist(), backend, stats)
    kept[BBOX_COL] = pd.Series(texts, index=kept.index, dtype=object)
    kept["width"] = widths
    kept["height"] = heights
    return kept, excluded


_LATE_FALLBACK = object()


def _replace_csv_core(input_csv_path, backend, fuse=None):
    """Shared front of the CSV -> CSV fast paths: native read, one native scan, the device stage, native emit.
    ``fuse`` = (min_boxes, iou_threshold) runs the fused K1+K2 launch and also yields the HIGH flag per table row
    (reference chain :262-281 -> :341-376); None runs K1 alone.  Returns NotImplemented when the fast path does not apply."""
    try:
        table = _fc.read_split(str(input_csv_path), [ANNOTATION_COL])
    except (OSError, ValueError, pd.errors.ParserError, UnicodeDecodeError):
        return NotImplemented
    if table is None or ANNOTATION_COL not in table.heavy:
        return NotImplemented
    be = _backend(backend)
    ann = table.heavy[ANNOTATION_COL]
    print(f"成功读取CSV，共 {table.n_rows} 行数据")
    kept_rows = np.flatnonzero(ann.na == 0)
    excluded_rows = np.flatnonzero(ann.na != 0)
    totals = {"boxes": 0, "points": 0, "host_boxes": 0, "python_cells": 0, "host_rows": 0}
    high = None
    if fuse is not None and _native_pipeline(be):
        # all-native pass: every worker thread scans, launches the fused kernel and emits its share of the cells
        scan = _nj.replace_iou_buffers(ann.data, ann.off, ann.na, fuse[0], fuse[1])
        irregular = np.flatnonzero(scan.status == _nj.IRREGULAR)
        py = (_replace_cells_python(ann.cells(irregular), be, totals) if len(irregular) else ([], [], []))   # may raise, like the reference
        high = scan.high.copy()
        text, off = scan.text_buffers()
    else:
        scan = _nj.scan_polygons_buffers(ann.data, ann.off, ann.na)
        irregular = np.flatnonzero(scan.status == _nj.IRREGULAR)
        py = (_replace_cells_python(ann.cells(irregular), be, totals) if len(irregular) else ([], [], []))   # may raise, like the reference
        if fuse is not None:
            arg4, high = be.bbox_iou_fused(scan.xy, scan.pt_off, scan.cell_box_off, fuse[0], fuse[1])
            high = high.astype(bool)
        elif scan.n_boxes:
            _, arg4 = be.bbox_minmax(scan.xy, scan.pt_off)
        else:
            arg4 = np.zeros((0, 4), np.int32)
        text, off = scan.emit_buffers(arg4)
    # width / height of the kept rows: numpy columns when every kept cell is plain (ints / floats / nothing), else the per-cell
    # lists whose dtype pandas infers like the reference's `filtered_df["width"] = [...]` (:295-296)
    plain_wh = len(irregular) == 0 and len(kept_rows) > 0 and not (scan.w_kind[kept_rows] == 3).any() \
        and not (scan.h_kind[kept_rows] == 3).any() and scan.w_kind[kept_rows].any() and scan.h_kind[kept_rows].any()
    if not plain_wh:
        widths, heights = scan.width_height(0), scan.width_height(1)
        for col, key in ((widths, "width"), (heights, "height")):
            for i, v in enumerate(col):
                if v is Ellipsis:
                    col[i] = json.loads(ann.cell(i)).get(key)
    new_na = (scan.status != _nj.OK).astype(np.uint8)
    if fuse is not None:
        high[scan.status != _nj.OK] = False               # no bbox text -> a NaN cell -> no boxes (:344-345)
        raw = None
        for i in np.flatnonzero((scan.iou_host != 0) & (scan.status == _nj.OK)).tolist():      # ints beyond 2^25: CPython decides
            raw = bytes(text) if raw is None else raw
            high[i] = _iou_mask_python([raw[off[i]:off[i + 1]].decode("utf-8")], fuse[0], fuse[1], be, totals)[0]
        for j, i in enumerate(irregular.tolist()):
            high[i] = _iou_mask_python([py[0][j]], fuse[0], fuse[1], be, totals)[0]
    if len(irregular):                                   # splice the Python-path results into the column
        cells = [None] * table.n_rows
        ok_rows = np.flatnonzero(scan.status == _nj.OK)
        raw = bytes(text)
        for i in ok_rows.tolist():
            cells[i] = raw[off[i]:off[i + 1]].decode("utf-8")
        for j, i in enumerate(irregular.tolist()):
            cells[i], widths[i], heights[i] = py[0][j], py[1][j], py[2][j]
        spec = _fc._series_column(pd.Series(cells, dtype=object))
        new_col = _fc.Utf8Column(spec[1], spec[2], spec[3])
    else:
        new_col = _fc.Utf8Column(text, off, new_na, scan)
    if plain_wh:
        def _col(kind, val):
            kind, val = kind[kept_rows], val[kept_rows]
            if (kind == 1).all():
                return pd.Series(val.astype(np.int64))
            out = val.copy()
            out[kind == 0] = np.nan
            return pd.Series(out)
        kw, kh = _col(scan.w_kind, scan.w_val), _col(scan.h_kind, scan.h_val)
    else:
        kw = pd.Series([widths[i] for i in kept_rows.tolist()])     # dtype inference of `kept["width"] = list` (:295)
        kh = pd.Series([heights[i] for i in kept_rows.tolist()])
    full_w = pd.Series(np.full(table.n_rows, np.nan, dtype=object) if kw.dtype == object else np.zeros(table.n_rows, kw.dtype))
    full_h = pd.Series(np.full(table.n_rows, np.nan, dtype=object) if kh.dtype == object else np.zeros(table.n_rows, kh.dtype))
    full_w.iloc[kept_rows] = kw.to_numpy()
    full_h.iloc[kept_rows] = kh.to_numpy()
    names, columns = [], []
    if "source" in table.light.columns:
        names.append("source"); columns.append(table.light["source"])
    names += [ANNOTATION_COL, BBOX_COL]
    columns += [ann, new_col]
    names += ["width", "height"]
    columns += [full_w, full_h]
    return {"table": table, "scan": scan, "kept_rows": kept_rows, "excluded_rows": excluded_rows, "names": names,
            "columns": columns, "high": high, "totals": totals}


def _replace_csv_write(core, output_csv_path, excluded_output_file, also=None, reread=False):
    """processed CSV (native writer) + excluded CSV; returns the step's result dict, or _LATE_FALLBACK when the writer's
    sample check against pandas refused the table (nothing written then).

    The processed file is written by a thread of its own, and what the IoU step needs is prepared meanwhile: ``reread`` leaves
    the light columns as that step's read_csv would type them in core["reread"] (_as_reread: pandas, 0.3 s per 300 k rows);
    ``also`` = [path, names, None, n_rows, rows] entries of the IoU step's two files (fused twin) are checked against pandas and
    written side by side with the processed file — entry[2] is then set to the columns; left None when the writer refused one
    of them (nothing of those files written).  A buffered write holds its file's inode lock, so the writer's threads take turns
    inside ONE file, while different files proceed in parallel."""
    table, kept_rows, excluded_rows = core["table"], core["kept_rows"], core["excluded_rows"]
    main = _fc.prepare_write(str(output_csv_path), core["names"], core["columns"], table.n_rows, kept_rows)
    if main is None:
        return _LATE_FALLBACK                              # the row count was already printed
    writing = [_fc.WriteInBackground(main)]
    try:
        if also or reread:
            try:
                cols = _as_reread(core, core["names"])
            except Exception:  # noqa: BLE001 - work done ahead for the IoU step: its failure belongs to that step
                cols = None
            if reread and cols is not None:
                core["reread"] = cols
            if also and cols is not None:
                jobs = [_fc.prepare_write(p_, nm_, cols, n_, r_) for p_, nm_, _, n_, r_ in also]
                if all(j is not None for j in jobs):
                    writing += [_fc.WriteInBackground(j) for j in jobs]
                    also[:] = [(p_, nm_, cols, n_, r_) for p_, nm_, _, n_, r_ in also]
        if excluded_output_file is not None:
            excluded = table.light.iloc[excluded_rows].copy()
            excluded.insert(table.names.index(ANNOTATION_COL), ANNOTATION_COL, np.nan)
            Path(excluded_output_file).parent.mkdir(parents=True, exist_ok=True)
            excluded[table.names].to_csv(excluded_output_file, index=False, encoding="utf-8-sig")
    finally:
        ok = [w.done() for w in writing]
    if not ok[0]:
        return _LATE_FALLBACK
    if also and len(ok) > 1 and not all(ok[1:]):
        also[:] = [(p_, nm_, None, n_, r_) for p_, nm_, _, n_, r_ in also]      # an I/O failure there: the caller's other route
    return {
        "filtered_rows": int(len(kept_rows)),
        "excluded_rows": int(len(excluded_rows)),
        "excluded_output": excluded_output_file,
    }


def _replace_csv_fast(input_csv_path, output_csv_path, excluded_output_file, backend):
    """CSV -> CSV replace step without pandas touching the annotation column (fastcsv + native JSON).
    Returns NotImplemented whenever the fast path does not apply; nothing has been written then.
    The pass also computes the IoU step's flag and parks the table for it (see _STEP_CACHE)."""
    clear_step_cache()
    core, fuse = NotImplemented, None
    if _step_cache_limit() > 0:
        fuse = _STEP_CACHE["params"]
        try:
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()) as quiet:
                core = _replace_csv_core(input_csv_path, backend, fuse=fuse)
            if core is not NotImplemented:
                print(quiet.getvalue(), end="")
        except Exception:  # noqa: BLE001  the IoU step's own failure (string coordinates ...) must not surface one step early
            core, fuse = NotImplemented, None
    if core is NotImplemented:
        fuse = None
        core = _replace_csv_core(input_csv_path, backend)
    if core is NotImplemented:
        return NotImplemented
    parked = False
    try:
        res = _replace_csv_write(core, output_csv_path, excluded_output_file, reread=fuse is not None)
        if res is not _LATE_FALLBACK and fuse is not None:
            heavy = [c for c in core["columns"] if isinstance(c, _fc.Utf8Column)]
            heavy_ok = all((c.na != 0).sum() < len(c) or len(c) == 0 for c in heavy)     # an all-NaN text column is re-read as float
            size = sum(int(c.off[-1]) for c in heavy)
            key = _file_key(output_csv_path) if heavy_ok and size <= _step_cache_limit() else None
            if key is not None:
                with _STEP_CACHE["lock"]:
                    _STEP_CACHE["entry"] = {"key": key, "params": fuse, "core": core}
                parked = True
        return res
    finally:
        if not parked:
            core["scan"].close()


def _iou_csv_cached(input_csv_path, high_iou_csv, other_csv, min_boxes, iou_threshold) -> bool:
    """the IoU step from the table the replace step parked (see _STEP_CACHE); False: not applicable, nothing written"""
    with _STEP_CACHE["lock"]:
        entry = _STEP_CACHE["entry"]
        if entry is None:
            return False
        if entry["params"] != (min_boxes, iou_threshold) or entry["key"][0] != os.path.abspath(str(input_csv_path)):
            return False
        _STEP_CACHE["entry"] = None                       # one use: the caller owns it now
    core = entry["core"]
    try:
        if _file_key(input_csv_path) != entry["key"]:     # the file was touched since: what is parked is not what is on disk
            return False
        kept_rows, high = core["kept_rows"], core["high"]
        cols = core.get("reread") or _as_reread(core, core["names"])      # (made while the replace step was writing its file)
        n = core["table"].n_rows
        return bool(_fc.write_tables([(str(high_iou_csv), core["names"], cols, n, kept_rows[high[kept_rows]]),
                                      (str(other_csv), core["names"], cols, n, kept_rows[~high[kept_rows]])]))
    finally:
        core["scan"].close()


def process_csv_replace_ptlist(
        input_csv_path: str,
        output_csv_path: str = "processed_replaced_ptlist.csv",
        excluded_output_file: Optional[str] = "processed_excluded.csv",
        backend=None,
):
    announced = False
    if _fc.enabled() and _nj.enabled() and os.path.isfile(str(input_csv_path)):
        res = _replace_csv_fast(input_csv_path, output_csv_path, excluded_output_file, backend)
        if res is _LATE_FALLBACK:
            announced = True
        elif res is not NotImplemented:
            LAST_IO_PATH["replace"] = "native"
            return res
    LAST_IO_PATH["replace"] = "pandas"
    try:
        df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
        if not announced:
            print(f"成功读取CSV，共 {len(df)} 行数据")
    except FileNotFoundError:
        print(f"错误：未找到文件 {input_csv_path}")
        return None
    except Exception as e:
        print(f"读取失败：{e}")
        return None
    if ANNOTATION_COL not in df.columns:
        print(f"错误：CSV缺少列 '{ANNOTATION_COL}'")
        return None

    kept, excluded = replace_ptlist_frame(df, backend)
    wanted = ["source", ANNOTATION_COL, BBOX_COL, "width", "height"]       # :298-306
    Path(output_csv_path).parent.mkdir(parents=True, exist_ok=True)
    kept[[c for c in wanted if c in kept.columns]].to_csv(output_csv_path, index=False, encoding="utf-8-sig")
    if excluded_output_file is not None:
        Path(excluded_output_file).parent.mkdir(parents=True, exist_ok=True)
        excluded.to_csv(excluded_output_file, index=False, encoding="utf-8-sig")
    return {
        "filtered_rows": len(kept),
        "excluded_rows": len(excluded),
        "excluded_output": excluded_output_file,
    }


# =============================================================================== a4  IoU filter
def _iou_mask_python(cells, min_boxes, iou_threshold, be, totals) -> np.ndarray:
    """flatten.py path for the IoU step (see _replace_cells_python)."""
    out = np.zeros(len(cells), bool)
    for start in range(0, len(cells), _CHUNK_CELLS):
        batch = _fl.flatten_boxes(cells[start:start + _CHUNK_CELLS])
        n = len(batch.row_off) - 1
        if n:
            out[start:start + n] = be.iou_any_ge(batch.box4, batch.row_off, min_boxes, iou_threshold).astype(bool)
        for ri, boxes in batch.host_rows.items():
            out[start + ri] = _fl.host_row_is_high(boxes, min_boxes, iou_threshold)
        totals["boxes"] += batch.stats["boxes"]
        totals["host_rows"] += batch.stats["host_rows"]
    return out


def _iou_mask_native(cells, min_boxes, iou_threshold, be, totals) -> np.ndarray:
    try:
        scan = _nj.scan_boxes(cells)
    except UnicodeEncodeError:
        totals["python_cells"] += len(cells)
        return _iou_mask_python(cells, min_boxes, iou_threshold, be, totals)
    irregular = np.flatnonzero(scan.status == _nj.IRREGULAR)
    totals["python_cells"] += int(len(irregular))
    py = (_iou_mask_python([cells[i] for i in irregular.tolist()], min_boxes, iou_threshold, be, totals)
          if len(irregular) else np.zeros(0, bool))
    out = be.iou_any_ge(scan.box4, scan.row_off, min_boxes, iou_threshold).astype(bool)
    out[irregular] = py
    totals["boxes"] += int(scan.row_off[-1])
    scan.close()
    return out


def iou_high_mask(cells, min_boxes: int = 2, iou_threshold: float = 0.98, backend=None,
                  stats: Optional[dict] = None) -> np.ndarray:
    """HIGH flag per bbox-JSON cell (:392-398): flatten -> K2 (native scanner for regular cells,
    flatten.py for the irregular ones), in batches of _NATIVE_CHUNK_CELLS cells."""
    be = _backend(backend)
    cells = list(cells)
    totals = {"rows": len(cells), "boxes": 0, "host_rows": 0, "python_cells": 0}
    if _nj.enabled():
        parts = [_iou_mask_native(cells[s:s + _NATIVE_CHUNK_CELLS], min_boxes, iou_threshold, be, totals)
                 for s in range(0, len(cells), _NATIVE_CHUNK_CELLS)]
        out = np.concatenate(parts) if parts else np.zeros(0, bool)
    else:
        totals["python_cells"] = len(cells)
        out = _iou_mask_python(cells, min_boxes, iou_threshold, be, totals)
    if stats is not None:
        stats.update(totals)
    return out


def iou_filter_frame(df: pd.DataFrame, min_boxes: int = 2, iou_threshold: float = 0.98, backend=None,
                     stats: Optional[dict] = None):
    """In-memory twin of the IoU step -> (high frame, other frame), rows in original order."""
    mask = iou_high_mask(df[BBOX_COL].tolist(), min_boxes, iou_threshold, backend, stats)
    return df[mask], df[~mask]


def _iou_csv_fast(input_csv_path, high_iou_csv, other_csv, min_boxes, iou_threshold, backend):
    """CSV -> two CSVs IoU step on flat buffers (see _replace_csv_fast)."""
    try:
        table = _fc.read_split(str(input_csv_path), [ANNOTATION_COL, BBOX_COL])
    except (OSError, ValueError, pd.errors.ParserError, UnicodeDecodeError):
        return NotImplemented
    if table is None or BBOX_COL not in table.heavy:
        return NotImplemented
    be = _backend(backend)
    col = table.heavy[BBOX_COL]
    scan = _nj.scan_boxes_buffers(col.data, col.off, col.na)
    totals = {"boxes": 0, "host_rows": 0, "python_cells": 0}
    irregular = np.flatnonzero(scan.status == _nj.IRREGULAR)
    py = (_iou_mask_python(col.cells(irregular), min_boxes, iou_threshold, be, totals) if len(irregular)
          else np.zeros(0, bool))                          # may raise, like the reference
    mask = be.iou_any_ge(scan.box4, scan.row_off, min_boxes, iou_threshold).astype(bool)
    mask[irregular] = py
    scan.close()
    columns = [table.heavy[nm] if nm in table.heavy else table.light[nm] for nm in table.names]
    # sample-check both files before writing either, so a fallback never leaves half the output behind
    ok = _fc.write_tables([(str(high_iou_csv), table.names, columns, table.n_rows, np.flatnonzero(mask)),
                           (str(other_csv), table.names, columns, table.n_rows, np.flatnonzero(~mask))])
    return None if ok else NotImplemented


def filter_by_box_count_and_iou(
        input_csv_path,
        high_iou_csv="high_iou_0.98.csv",
        other_csv="other_data.csv",
        min_boxes: int = 2,
        iou_threshold: float = 0.98,
        backend=None,
):
    _STEP_CACHE["params"] = (min_boxes, iou_threshold)              # what the next replace step computes ahead
    if _fc.enabled() and _nj.enabled() and os.path.isfile(str(input_csv_path)):
        if _iou_csv_cached(input_csv_path, high_iou_csv, other_csv, min_boxes, iou_threshold):
            LAST_IO_PATH["iou"] = "cached"
            return
        if _iou_csv_fast(input_csv_path, high_iou_csv, other_csv, min_boxes, iou_threshold, backend) is None:
            LAST_IO_PATH["iou"] = "native"
            return
    LAST_IO_PATH["iou"] = "pandas"
    try:
        df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
    except Exception as e:
        print(f"读取失败：{e}")
        return
    if BBOX_COL not in df.columns:
        print(f"错误：缺少必要列 {BBOX_COL}")
        return
    high, other = iou_filter_frame(df, min_boxes, iou_threshold, backend)
    Path(high_iou_csv).parent.mkdir(parents=True, exist_ok=True)
    Path(other_csv).parent.mkdir(parents=True, exist_ok=True)
    high.to_csv(high_iou_csv, index=False, encoding="utf-8-sig")
    other.to_csv(other_csv, index=False, encoding="utf-8-sig")


# =============================================================================== a3 + a4  replace -> IoU in one pass
# The processing page runs the two steps back to back on the same rows (reference ui/pages/processing.py:580-598), and the
# replace step's output box IS the two-point ptList the IoU step reads back (:260 -> :354-362).  The functions below do both
# with ONE native scan, ONE fused K1+K2 launch (dyd_bbox_iou_fused, points / offsets resident on the device between the
# two stages) and one native emit.  Results are those of the two reference steps in sequence, including the row whose
# polygon has no valid point: it is emitted with null coordinates and ends the row's IoU box list (:254-255, :364-365).
def _replace_iou_cells_native(cells, min_boxes, iou_threshold, be, totals):
    """one batch: native scan -> fused K1+K2 -> native emit.  -> (texts object array, widths, heights, high bool array);
    widths / heights are numpy columns when every cell is plain (PolygonScan.wh_column), else per-cell lists"""
    import time as _t
    t0 = _t.perf_counter()
    if _native_pipeline(be):
        res = _replace_iou_cells_pipeline(cells, min_boxes, iou_threshold, be, totals)
        if res is not None:
            return res
    try:
        scan = _nj.scan_polygons(cells)
    except UnicodeEncodeError:                         # a lone surrogate somewhere: the two steps in sequence, CPython flatten
        totals["python_cells"] += len(cells)
        texts, widths, heights = _replace_cells_python(list(cells), be, totals)
        arr = np.empty(len(texts), object)
        arr[:] = texts
        return arr, widths, heights, _iou_mask_python(texts, min_boxes, iou_threshold, be, totals)
    t1 = _t.perf_counter()
    irregular = np.flatnonzero(scan.status == _nj.IRREGULAR)
    totals["python_cells"] += int(len(irregular))
    # irregular cells first: they are the only ones that can raise, and they must raise before any output
    py = _replace_cells_python([cells[i] for i in irregular.tolist()], be, totals) if len(irregular) else ([], [], [])
    t2 = _t.perf_counter()
    arg4, high = be.bbox_iou_fused(scan.xy, scan.pt_off, scan.cell_box_off, min_boxes, iou_threshold)
    t3 = _t.perf_counter()
    high = high.astype(bool)
    texts = scan.emit_array(arg4)
    t4 = _t.perf_counter()
    for k, v in (("s_scan", t1 - t0), ("s_python_cells", t2 - t1), ("s_device", t3 - t2), ("s_emit", t4 - t3)):
        totals[k] = totals.get(k, 0.0) + v
    plain = len(irregular) == 0
    widths, heights = (scan.wh_column(0), scan.wh_column(1)) if plain else (scan.width_height(0), scan.width_height(1))
    for col, key in ((widths, "width"), (heights, "height")):
        if isinstance(col, list):
            for i, v in enumerate(col):
                if v is Ellipsis:                      # rare value kinds (str / container / huge int): ask CPython
                    col[i] = json.loads(cells[i]).get(key)
    high[scan.status != _nj.OK] = False                # no bbox text -> a NaN cell -> no boxes (:344-345)
    for i in np.flatnonzero((scan.iou_host != 0) & (scan.status == _nj.OK)).tolist():   # ints beyond 2^25: CPython decides
        high[i] = _iou_mask_python([texts[i]], min_boxes, iou_threshold, be, totals)[0]
    for j, i in enumerate(irregular.tolist()):
        texts[i], widths[i], heights[i] = py[0][j], py[1][j], py[2][j]
        high[i] = _iou_mask_python([py[0][j]], min_boxes, iou_threshold, be, totals)[0]
    totals["boxes"] += scan.n_boxes
    totals["points"] += int(scan.xy.shape[0])
    totals["fused_launches"] += 1
    totals["fast_cells"] += scan.fast_cells
    t5 = _t.perf_counter()
    scan.close()
    totals["s_fixups"] = totals.get("s_fixups", 0.0) + (t5 - t4)
    totals["s_release"] = totals.get("s_release", 0.0) + (_t.perf_counter() - t5)
    return texts, widths, heights, high


def _native_pipeline(be) -> bool:
    """the all-native replace -> IoU pass applies when the device stage is the product's own (not an injected checker)"""
    from .. import _native as _nat
    return be is _nat and os.environ.get("DYD_NATIVE_PIPELINE", "1") != "0"


def _replace_iou_cells_pipeline(cells, min_boxes, iou_threshold, be, totals, arrow: bool = False):
    """one batch through dyd_json_replace_iou: every worker thread scans its share of the cells, launches the fused kernel on its
    own arrays and emits — no gathered copies.  Returns None when the cells cannot be viewed (lone surrogate): the caller's
    stepwise route handles that."""
    import time as _t
    t0 = _t.perf_counter()
    try:
        r = _nj.replace_iou(cells, min_boxes, iou_threshold)
    except UnicodeEncodeError:
        return None
    t1 = _t.perf_counter()
    irregular = np.flatnonzero(r.status == _nj.IRREGULAR)
    totals["python_cells"] += int(len(irregular))
    # irregular cells: the only ones that can raise; nothing has been handed out yet
    py = _replace_cells_python([cells[i] for i in irregular.tolist()], be, totals) if len(irregular) else ([], [], [])
    t2 = _t.perf_counter()
    needs_objects = len(irregular) > 0 or bool(((r.iou_host != 0) & (r.status == _nj.OK)).any())
    as_arrow = arrow and not needs_objects          # cells the host must patch or re-read need str objects
    texts = r.texts_arrow() if as_arrow else r.texts_array()
    t3 = _t.perf_counter()
    high = r.high.copy()
    plain = len(irregular) == 0
    widths, heights = (r.wh_column(0), r.wh_column(1)) if plain else (r.width_height(0), r.width_height(1))
    for col, key in ((widths, "width"), (heights, "height")):
        if isinstance(col, list):
            for i, v in enumerate(col):
                if v is Ellipsis:
                    col[i] = json.loads(cells[i]).get(key)
    high[r.status != _nj.OK] = False
    for i in np.flatnonzero((r.iou_host != 0) & (r.status == _nj.OK)).tolist():
        high[i] = _iou_mask_python([texts[i]], min_boxes, iou_threshold, be, totals)[0]
    for j, i in enumerate(irregular.tolist()):
        texts[i], widths[i], heights[i] = py[0][j], py[1][j], py[2][j]
        high[i] = _iou_mask_python([py[0][j]], min_boxes, iou_threshold, be, totals)[0]
    totals["boxes"] += r.n_boxes
    totals["points"] += r.n_points
    totals["fused_launches"] += r.n_parts
    totals["fast_cells"] += r.fast_cells
    t4 = _t.perf_counter()
    if not as_arrow:
        r.close()                                  # (an Arrow column lives on the handle's buffers and keeps it alive)
    for k, v in (("s_pipeline", t1 - t0), ("s_python_cells", t2 - t1), ("s_strings", t3 - t2), ("s_fixups", t4 - t3),
                 ("s_release", _t.perf_counter() - t4), ("s_part_scan", r.seconds["scan"]), ("s_part_device", r.seconds["device"]),
                 ("s_part_emit", r.seconds["emit"])):
        totals[k] = totals.get(k, 0.0) + v
    totals["native_pipeline"] = totals.get("native_pipeline", 0) + 1
    return texts, widths, heights, high


def _join_columns(parts):
    """per-batch column values (numpy arrays or lists) -> one value for ``frame[col] = ...``"""
    if len(parts) == 1:
        return parts[0]
    if all(isinstance(p, np.ndarray) for p in parts) and len({p.dtype for p in parts}) == 1:
        return np.concatenate(parts)
    out = []
    for p in parts:
        out.extend(p.tolist() if isinstance(p, np.ndarray) else p)
    return out


def _replace_and_filter_arrays(cells, min_boxes, iou_threshold, be, totals, arrow: bool = False):
    """cells: object ndarray / list.  -> (texts object array — or a pandas ArrowStringArray when `arrow` and one native batch
    without host-decided cells covers the column —, widths, heights, high) over all batches"""
    if not _nj.enabled():                              # DYD_NATIVE_JSON=0: the two steps in sequence on the CPython flatten
        totals["python_cells"] = len(cells)
        texts, widths, heights = _replace_cells_python(list(cells), be, totals)
        arr = np.empty(len(texts), object)
        arr[:] = texts
        return arr, widths, heights, _iou_mask_python(texts, min_boxes, iou_threshold, be, totals)
    if arrow and _native_pipeline(be) and 0 < len(cells) <= _NATIVE_CHUNK_CELLS:
        res = _replace_iou_cells_pipeline(cells, min_boxes, iou_threshold, be, totals, arrow=True)
        if res is not None:
            return res
    t_p, w_p, h_p, m_p = [], [], [], []
    for start in range(0, len(cells), _NATIVE_CHUNK_CELLS):
        t, w, h, m = _replace_iou_cells_native(cells[start:start + _NATIVE_CHUNK_CELLS], min_boxes, iou_threshold, be, totals)
        t_p.append(t); w_p.append(w); h_p.append(h); m_p.append(m)
    if not t_p:
        return np.empty(0, object), [], [], np.zeros(0, bool)
    return np.concatenate(t_p), _join_columns(w_p), _join_columns(h_p), np.concatenate(m_p)


def replace_and_filter_cells(cells, min_boxes: int = 2, iou_threshold: float = 0.98, backend=None,
                             stats: Optional[dict] = None) -> tuple:
    """(new JSON text or None, width, height) per annotation cell plus the HIGH flag the IoU step would give the row:
    native scan -> fused K1+K2 -> native emit, in batches of _NATIVE_CHUNK_CELLS cells."""
    be = _backend(backend)
    cells = list(cells)
    totals = {"cells": len(cells), "boxes": 0, "points": 0, "host_boxes": 0, "host_rows": 0, "python_cells": 0,
              "fused_launches": 0, "fast_cells": 0}
    texts, widths, heights, high = _replace_and_filter_arrays(cells, min_boxes, iou_threshold, be, totals)
    if stats is not None:
        stats.update(totals)
    return (texts.tolist(), widths.tolist() if isinstance(widths, np.ndarray) else widths,
            heights.tolist() if isinstance(heights, np.ndarray) else heights, high)


def replace_and_filter_frame(df: pd.DataFrame, min_boxes: int = 2, iou_threshold: float = 0.98, backend=None,
                             stats: Optional[dict] = None, text_dtype: str = "object"):
    """In-memory twin of replace_ptlist -> iou_filter run back to back:
    -> (kept frame with the three new columns, excluded rows, HIGH rows of kept, other rows of kept).
    The annotation cells are read in place (UTF-8 views of the column's str objects) and the new column's str objects are
    created natively, so no per-cell Python work remains for regular cells.  ``text_dtype="arrow"`` returns the new bbox column
    as pandas' Arrow-backed ``string`` dtype laid directly over the emitter's buffers (no str objects at all; same values,
    missing cells are ``pd.NA`` instead of ``None``) — the default keeps the reference's object column of str."""
    import time as _t
    be = _backend(backend)
    t0 = _t.perf_counter()
    from .. import pycells
    col = df[ANNOTATION_COL]
    if len(df) >= 65536 and col.dtype == object and pycells.all_str(col.to_numpy()):
        # a column of str cells: nothing to drop (isna walks a million objects to say so); the same frames as the masks below give
        kept, excluded = df.copy(), df[np.zeros(len(df), dtype=bool)].copy()
    else:
        na = col.isna()
        kept = df[~na].copy()                                      # == dropna(subset=[col]).copy() (:249)
        excluded = df[na].copy()                                   # :250
    totals = {"cells": len(kept), "boxes": 0, "points": 0, "host_boxes": 0, "host_rows": 0, "python_cells": 0,
              "fused_launches": 0, "fast_cells": 0}
    t1 = _t.perf_counter()
    if text_dtype not in ("object", "arrow"):
        raise ValueError('text_dtype must be "object" or "arrow"')
    texts, widths, heights, high = _replace_and_filter_arrays(kept[ANNOTATION_COL].to_numpy(), min_boxes, iou_threshold, be, totals,
                                                              arrow=(text_dtype == "arrow"))
    t2 = _t.perf_counter()
    kept[BBOX_COL] = (pd.Series(texts, index=kept.index, dtype=object) if isinstance(texts, np.ndarray)
                      else pd.Series(texts, index=kept.index))
    kept["width"] = widths
    kept["height"] = heights
    out = (kept, excluded, _frame_rows(kept, high, keep_labels=True), _frame_rows(kept, ~high, keep_labels=True))
    totals["s_frame_in"] = t1 - t0
    totals["s_frame_out"] = _t.perf_counter() - t2
    if stats is not None:
        stats.update(totals)
    return out


def _as_reread(core, names):
    """The light columns of the processed table as the IoU step's read_csv would type them (:379): the same table width
    (heavy cells left empty), parsed by pandas itself; scattered back to table rows for the writer."""
    table, kept_rows = core["table"], core["kept_rows"]
    light = {nm: (col.iloc[kept_rows].reset_index(drop=True) if not isinstance(col, _fc.Utf8Column) else np.nan)
             for nm, col in zip(names, core["columns"])}
    text = pd.DataFrame(light, columns=names, index=pd.RangeIndex(len(kept_rows))).to_csv(index=False)
    light_names = [nm for nm, col in zip(names, core["columns"]) if not isinstance(col, _fc.Utf8Column)]
    back = pd.read_csv(io.StringIO(text), usecols=light_names) if light_names else pd.DataFrame()
    out = []
    for nm, col in zip(names, core["columns"]):
        if isinstance(col, _fc.Utf8Column):
            out.append(col)
            continue
        vals = back[nm].to_numpy()
        full = np.empty(table.n_rows, dtype=vals.dtype)
        if vals.dtype == object:
            full[:] = np.nan
        else:
            full[:] = 0
        full[kept_rows] = vals
        out.append(pd.Series(full))
    return out


def process_csv_replace_and_filter(
        input_csv_path: str,
        output_csv_path: str = "processed_replaced_ptlist.csv",
        excluded_output_file: Optional[str] = "processed_excluded.csv",
        high_iou_csv="high_iou_0.98.csv",
        other_csv="other_data.csv",
        min_boxes: int = 2,
        iou_threshold: float = 0.98,
        backend=None,
):
    """process_csv_replace_ptlist(input, output, excluded) followed by filter_by_box_count_and_iou(output, high, other,
    min_boxes, iou_threshold) — the same five files, prints and return value (the replace step's dict, or None), from one
    native read, one native scan, one fused K1+K2 launch and one native emit.  Whenever the fast path does not apply (or
    its writer refuses a table) the two step functions are simply called in sequence."""
    if _fc.enabled() and _nj.enabled() and os.path.isfile(str(input_csv_path)):
        core = _replace_csv_core(input_csv_path, backend, fuse=(min_boxes, iou_threshold))
        if core is not NotImplemented:
            try:
                heavy_ok = all((c.na != 0).sum() < len(c) or len(c) == 0 for c in core["columns"]
                               if isinstance(c, _fc.Utf8Column))                    # an all-NaN text column is re-read as float
                kept_rows, high = core["kept_rows"], core["high"]
                n = core["table"].n_rows
                both = [(str(high_iou_csv), core["names"], None, n, kept_rows[high[kept_rows]]),
                        (str(other_csv), core["names"], None, n, kept_rows[~high[kept_rows]])]
                res = _replace_csv_write(core, output_csv_path, excluded_output_file, also=both) if heavy_ok else _LATE_FALLBACK
                if res is not _LATE_FALLBACK:
                    ok = both[0][2] is not None                # the three files went out side by side
                    if not ok:
                        cols = _as_reread(core, core["names"])
                        ok = _fc.write_tables([(p_, nm_, cols, n_, r_) for p_, nm_, _, n_, r_ in both])
                    if ok:
                        LAST_IO_PATH["replace_iou"] = "fused-native"
                        LAST_IO_PATH["replace"] = LAST_IO_PATH["iou"] = "native"
                        return res
                    LAST_IO_PATH["replace_iou"] = "fused-native + iou step"
                    LAST_IO_PATH["replace"] = "native"
                    filter_by_box_count_and_iou(output_csv_path, high_iou_csv, other_csv, min_boxes, iou_threshold, backend)
                    return res
            finally:
                core["scan"].close()
            # the writer refused the processed table after the row count was printed: pandas writes it, silently
            LAST_IO_PATH["replace_iou"] = "two steps"
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                res = process_csv_replace_ptlist(input_csv_path, output_csv_path, excluded_output_file, backend)
            filter_by_box_count_and_iou(output_csv_path, high_iou_csv, other_csv, min_boxes, iou_threshold, backend)
            return res
    LAST_IO_PATH["replace_iou"] = "two steps"
    res = process_csv_replace_ptlist(input_csv_path, output_csv_path, excluded_output_file, backend)
    if res is not None:
        filter_by_box_count_and_iou(output_csv_path, high_iou_csv, other_csv, min_boxes, iou_threshold, backend)
    return res


# =============================================================================== a5  split
def rules_to_label_map(rules_df: pd.DataFrame, rule_mode: str = "wide", label_col=None, category_col=None) -> dict:
    """label -> category from the rules sheet (:688-703); later entries overwrite earlier ones."""
    mapping = {}
    if rule_mode == "wide":
        for column in rules_df.columns:
            category = str(column).strip()
            if not category:
                continue
            for cell in rules_df[column].dropna():
                for label in _split_label_cell(cell):
                    mapping[label] = category
    elif rule_mode == "two_column":
        for _, rule in rules_df.iterrows():
            label = str(rule.get(label_col, "")).strip()
            category = str(rule.get(category_col, "")).strip()
            if label and category and label.lower() != "nan" and category.lower() != "nan":
                mapping[label] = category
    return mapping


def split_cut_sizes(n: int, train_ratio: float, val_ratio: float, test_ratio: float) -> tuple:
    """(n_train, n_val) = (int(n*tr), int(n*va)) after normalising the ratios by their sum (:673-676, :802-803)."""
    total = train_ratio + val_ratio + test_ratio
    train_ratio /= total
    val_ratio /= total
    return int(n * train_ratio), int(n * val_ratio)


_SPLIT_ERRORS = {1: "空数据", 2: "JSON解析失败", 3: "objects不是列表", 4: "标注字段objects为空"}      # utils.py:645-657, :722


def _expand_cell_python(cell, label_to_category):
    """One row of the split step the way the reference walks it (processor.py:720-792), for the cells the native
    expansion leaves to CPython.  -> (error or None, combo, [(label, json)], [(kind, label)], joined reasons)"""
    doc, objs, err = _parse_data_objects(cell)
    if err or not objs:
        return err or _SPLIT_ERRORS[4], "", [], [], ""
    seen = set()
    for o in objs:
        if isinstance(o, dict) and o.get("name"):
            seen.update(_split_object_labels(o.get("name")))
    combo = "，".join(sorted(seen)) if seen else ""
    rows, events, reasons = [], [], set()
    for o in objs:
        if not isinstance(o, dict):
            continue
        labels = _split_object_labels(o.get("name"))
        if not labels:
            events.append((_nj.EV_NO_NAME, None))
            continue
        for label in labels:
            if label not in label_to_category:
                events.append((_nj.EV_UNDEFINED, label))
                reasons.add(f"标签{label}未在规则中定义")
                continue
            single = dict(o)                                 # the reference deep-copies (:764); the JSON text is the same
            single["name"] = label
            slim = {k: v for k, v in doc.items() if k != "objects"}
            slim["objects"] = [single]
            rows.append((label, json.dumps(slim, ensure_ascii=False)))
    if not rows:
        events.append((_nj.EV_NOTHING_CLASSIFIED, None))
    return None, combo, rows, events, "；".join(sorted(reasons))


class _Expansion:
    """Every row of the split step expanded (:712-792): records (source row, label code, JSON text) in row order, the per-row
    bookkeeping of split_counts and the unclassified entries in the reference's append order.  The record texts are str objects
    only once somebody asks for them (``json_take``): at table scale they stay in the native handle until the frames are built,
    and are then created directly in shuffled order."""

    def json_take(self, idx=None) -> np.ndarray:
        """object array of the records' JSON text, record idx[i] at place i (all records in row order without idx)"""
        if self.json_objs is not None:
            return self.json_objs if idx is None else _pycells.take(self.json_objs, idx)
        if self.native is None or not self.n_records:
            return np.empty(0, object)
        return self.native.record_strings(idx)

    def json_at_slots(self, slot: np.ndarray, arrow: bool = False):
        """the records' JSON text with record e at place slot[e] (a permutation): an object array of str made in ONE row-ordered
        walk over the native buffers, or (``arrow``) one gathered utf-8 buffer + offsets [n+1] for Arrow string columns"""
        if arrow:
            if self.json_objs is None and self.native is not None and self.n_records:
                return self.native.record_text(None, slot)
            objs = self.json_take()
            raw = [t.encode("utf-8") for t in objs.tolist()]
            lens = np.zeros(len(raw), np.int64)
            lens[slot] = [len(r) for r in raw]
            off = np.zeros(len(raw) + 1, np.int64)
            np.cumsum(lens, out=off[1:])
            text = np.zeros(max(int(off[-1]), 1), np.uint8)
            for e, r in enumerate(raw):
                text[off[slot[e]]:off[slot[e]] + len(r)] = np.frombuffer(r, np.uint8)
            return text[:int(off[-1])], off
        if self.json_objs is not None:
            return _pycells.take(self.json_objs, None, slot=slot, checked=True)
        if self.native is None or not self.n_records:
            return np.empty(0, object)
        return self.native.record_strings(None, slot)

    def close(self):
        if self.native is not None:
            self.native.close()


SPLIT_BATCH_ROWS = 40_000       # tables of at least two such batches are expanded in up to 8 batches (native_json.SplitExpansionBatches)


def _expand_table(n: int, cell_of, label_to_category: dict, cells=None, views=None, strings: bool = False) -> _Expansion:
    """Expansion of all rows: native for the regular cells (csrc/host_json.cpp + host_split_fast.h), ``_expand_cell_python`` for
    the rest, merged back into row order.  ``views`` = (ptr, len, missing) of the picked cells (the DataFrame's own str objects),
    else ``cells`` is the list of picked cells; ``cell_of(i)`` returns row i's picked cell for the Python path.  ``strings``: the
    caller will ask for the records as str objects, so a large table's are allocated while later batches are still parsed."""
    labels = list(label_to_category)
    label_ix = {lab: i for i, lab in enumerate(labels)}
    ex = None
    if n and _nj.enabled():
        try:
            if views is not None and strings and n >= 2 * SPLIT_BATCH_ROWS:
                ex = _nj.split_expand_views_batched(*views, labels, n_batches=min(8, n // SPLIT_BATCH_ROWS))
            else:
                ex = _nj.split_expand_views(*views, labels) if views is not None else _nj.split_expand(cells, labels)
        except UnicodeEncodeError:                             # a lone surrogate somewhere: CPython handles every cell
            ex = None
    out = _Expansion()
    out.native, out.json_objs, out.labels = ex, None, labels
    if ex is not None:
        status = ex.status
        combo, reasons, n_out = ex.combo, ex.reasons, ex.n_expanded.astype(np.int64)
        has_reason = ex.reasons_nonempty.copy()
        src, code = ex.row_cell, ex.row_label
        e_src, e_kind, e_code, undef = ex.event_cell, ex.event_kind, ex.event_code, list(ex.undefined)
        python_rows = np.flatnonzero(status == _nj.SP_IRREGULAR).tolist()
    else:
        status = np.zeros(n, np.uint8)
        combo, reasons, n_out = np.full(n, "", object), np.full(n, "", object), np.zeros(n, np.int64)
        has_reason = np.zeros(n, bool)
        src, code = np.zeros(0, np.int64), np.zeros(0, np.int32)
        e_src, e_kind, e_code, undef = np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.int32), []
        python_rows = range(n)
    error = np.empty(n, object)                                # None everywhere
    for c, text in _SPLIT_ERRORS.items():
        hit = status == c
        if hit.any():
            error[hit] = text
    failed = (status >= 1) & (status <= 4)

    if len(python_rows):
        undef_ix = {lab: i for i, lab in enumerate(undef)}
        p_src, p_code, p_json, pe_src, pe_kind, pe_code = [], [], [], [], [], []
        for ri in python_rows:
            err, cmb, rows, events, why = _expand_cell_python(cell_of(ri), label_to_category)
            if err is not None:
                error[ri], failed[ri] = err, True
            combo[ri], reasons[ri], n_out[ri], has_reason[ri] = cmb, why, len(rows), bool(why)
            for lab, text in rows:
                p_src.append(ri); p_code.append(label_ix[lab]); p_json.append(text)
            for kind, lab in events:
                c = -1
                if kind == _nj.EV_UNDEFINED:
                    c = undef_ix.get(lab)
                    if c is None:
                        c = undef_ix[lab] = len(undef)
                        undef.append(lab)
                pe_src.append(ri); pe_kind.append(kind); pe_code.append(c)
        if p_src:
            texts = np.empty(len(p_json), object)
            texts[:] = p_json
            src = np.concatenate([src, np.asarray(p_src, np.int64)])
            order = np.argsort(src, kind="stable")
            src = src[order]
            code = np.concatenate([code, np.asarray(p_code, np.int32)])[order]
            out.json_objs = np.concatenate([ex.record_strings() if ex is not None and ex.n_records else np.empty(0, object), texts])[order]
        if pe_src:
            e_src = np.concatenate([e_src, np.asarray(pe_src, np.int64)])
            order = np.argsort(e_src, kind="stable")
            e_src = e_src[order]
            e_kind = np.concatenate([e_kind, np.asarray(pe_kind, np.uint8)])[order]
            e_code = np.concatenate([e_code, np.asarray(pe_code, np.int32)])[order]
    if ex is not None and out.json_objs is None:
        out.label_first, out.label_count = ex.label_stats(len(labels))
    else:
        out.label_count = np.bincount(code, minlength=len(labels)).astype(np.int64)
        out.label_first = np.full(len(labels), -1, np.int64)
        if len(code):
            uniq, first = np.unique(code, return_index=True)
            out.label_first[uniq] = first
    out.n_records, out.src_row, out.label_code = len(src), src, code

    # unclassified entries in the reference's append order: per row its events, an error row contributes itself
    err_rows = np.flatnonzero(failed)
    if len(err_rows):
        all_src = np.concatenate([e_src, err_rows])
        eorder = np.argsort(all_src, kind="stable")
        unc_row = all_src[eorder]
        unc_kind = np.concatenate([e_kind, np.zeros(len(err_rows), np.uint8)])[eorder]
        unc_code = np.concatenate([e_code, np.full(len(err_rows), -1, np.int32)])[eorder]
    else:
        unc_row, unc_kind, unc_code = e_src, e_kind, e_code
    # reason text per entry, built per distinct value rather than per entry: a table of the fixed texts and one text per
    # distinct undefined label, entries whose text is their row's own (errors, joined reasons) patched in afterwards
    is_err, is_undef, is_none = unc_kind == 0, unc_kind == _nj.EV_UNDEFINED, unc_kind == _nj.EV_NOTHING_CLASSIFIED
    table = np.empty(len(undef) + 2, object)
    table[0] = "标注框缺少name字段"                                                          # EV_NO_NAME (:747)
    table[1] = "标签无法匹配规则"                                                            # :779, a row without reasons
    table[2:] = [f"标签{lab}未在规则中定义" for lab in undef]                                 # :755
    tcode = np.where(is_undef, unc_code + 2, np.where(is_none, 1, 0)).astype(np.int32)
    unc_reason = _pycells.take_small(table, tcode)
    if is_err.any():
        unc_reason[is_err] = error[unc_row[is_err]]
    if is_none.any():
        rows_none = unc_row[is_none]
        named = has_reason[rows_none]
        if named.any():
            where = np.flatnonzero(is_none)[named]
            unc_reason[where] = reasons[rows_none[named]]
    undef_arr = np.empty(len(undef) + 1, object)                                            # [-1] stays None
    undef_arr[:len(undef)] = undef
    out.unc_row, out.unc_reason, out.unc_has_label = unc_row, unc_reason, is_undef
    out.unc_label = _pycells.take_small(undef_arr, np.where(is_undef, unc_code, len(undef)).astype(np.int32))   # None unless the entry names a label
    out.verdict_code = np.where(failed | (n_out == 0), 0, np.where(has_reason, 1, 2)).astype(np.int8)
    reasons_of_row = reasons
    if failed.any():
        reasons_of_row = reasons.copy()
        reasons_of_row[failed] = error[failed]                 # split_counts carries the error text there (:726)
    out.combo_of_row, out.n_out, out.reasons_of_row, out.failed = combo, n_out, reasons_of_row, failed
    return out


_VERDICTS = np.asarray(["否", "部分可分类", "是"], object)          # :782-784


def _expand_rows(cells, label_to_category: dict) -> dict:
    """The expansion over a plain list of picked cells, everything materialised (tests and small callers)."""
    cells = list(cells)
    ex = _expand_table(len(cells), cells.__getitem__, label_to_category, cells=cells)
    lab_arr = np.empty(len(ex.labels), object)
    lab_arr[:] = ex.labels
    res = {"src_row": ex.src_row, "label": lab_arr[ex.label_code] if ex.n_records else np.empty(0, object),
           "json": ex.json_take(), "combo_of_row": ex.combo_of_row, "n_out": ex.n_out, "verdict": _VERDICTS[ex.verdict_code],
           "reasons_of_row": ex.reasons_of_row, "unc_row": ex.unc_row, "unc_reason": ex.unc_reason, "unc_label": ex.unc_label}
    ex.close()
    return res


def _column_values(df: pd.DataFrame, name) -> np.ndarray:
    col = df[name]
    if isinstance(col, pd.DataFrame):                          # duplicated column name: the last one, as row[name] would be ambiguous
        col = col.iloc[:, -1]
    return col.to_numpy()


def _take_column(df: pd.DataFrame, name, idx: np.ndarray, slot=None, idx_at_slot=None):
    """out[slot[i]] = df[name].iloc[idx[i]] as an array for DataFrame(dict): object columns and plain numeric ones through the
    threaded builders (idx is walked in order: pass it sorted), extension arrays through their own take (``idx_at_slot`` =
    idx already permuted, made once by the caller)"""
    col = df[name]
    if isinstance(col, pd.DataFrame):
        col = col.iloc[:, -1]
    arr = col.array
    if isinstance(arr, pd.arrays.NumpyExtensionArray) or isinstance(col.dtype, np.dtype):
        return _pycells.take(col.to_numpy(), idx, checked=True, slot=slot)
    return arr.take(idx if slot is None else idx_at_slot())


def _picked_cells(df: pd.DataFrame, json_columns: list):
    """Per row the first non-empty str among the JSON columns (:713-718).  -> (views or None, cells list or None, cell_of)"""
    n = len(df)
    present = [c for c in json_columns if c in df.columns]
    arrays = [np.asarray(_column_values(df, c), dtype=object) for c in present]
    if n and _pycells.available() and _nj.enabled():
        ptr, length, missing = np.zeros(n, np.uint64), np.zeros(n, np.int64), np.ones(n, np.uint8)
        which = np.full(n, -1, np.int8)
        for k, arr in enumerate(arrays):
            v = _pycells.CellViews(arr)
            use = (missing != 0) & (v.missing == 0) & (v.len > 0)
            if k == 0 and use.all():
                ptr, length, missing, which = v.ptr, v.len, v.missing, np.zeros(n, np.int8)
                break
            ptr[use], length[use], missing[use], which[use] = v.ptr[use], v.len[use], 0, k
        def cell_of(i):
            return arrays[which[i]][i] if which[i] >= 0 else None
        return (ptr, length, missing), None, cell_of, arrays          # arrays keep the str objects alive
    cells = [None] * n
    for arr in reversed(arrays):
        for ri in range(n):
            v = arr[ri]
            if isinstance(v, str) and v:
                cells[ri] = v
    return None, cells, cells.__getitem__, arrays


def split_frames(df: pd.DataFrame, label_to_category: dict, json_columns: Optional[list] = None,
                 train_ratio: float = 0.8, val_ratio: float = 0.1, test_ratio: float = 0.1,
                 random_seed: int = 42, backend=None, stats: Optional[dict] = None, text_dtype: str = "object") -> dict:
    """In-memory twin of the split step (no Excel I/O).

    Host: expand every row into one record per (object, label in the rules), in object-then-label
    order (:741-775) — natively, straight from the DataFrame's str objects.  Device: K8 + K6 rank each record inside its
    category, apply the MT19937 permutation of ``sample(frac=1, random_state=seed)`` and assign train/val/test (:800-806).
    Host: per category ONE gathered take per column in shuffled order (the record texts become str objects right there), and
    the three sheets are slices of that frame, as in the reference (:804-806).

    ``text_dtype="arrow"`` returns the JSON columns of the category frames as pandas' Arrow-backed ``string`` dtype over one
    gathered buffer per category (same values, no str objects); the default keeps the reference's object columns of str.

    -> {"categories": {cat: (train, val, test)}, "unclassified": frame, "split_counts": frame,
        "category_counts": {cat: n}, "expanded": {src_row, category_id, position, split}}"""
    import time as _time

    if text_dtype not in ("object", "arrow"):
        raise ValueError('text_dtype must be "object" or "arrow"')
    be = _backend(backend)
    t0 = _time.perf_counter()
    if json_columns is None:                                        # :680-685
        json_columns = [c for c in (BBOX_COL, ANNOTATION_COL) if c in df.columns]
    present_json = [c for c in json_columns if c in df.columns]
    cols = list(df.columns)
    n = len(df)

    views, cells, cell_of, keep_alive = _picked_cells(df, json_columns)
    ex = _expand_table(n, cell_of, label_to_category, cells=cells, views=views, strings=text_dtype == "object")
    t1 = _time.perf_counter()

    # ---- categories in first-appearance order (:773, dict insertion order); category id per record -----------------
    labels = ex.labels
    cat_first = {}
    for li in np.argsort(np.where(ex.label_first < 0, np.iinfo(np.int64).max, ex.label_first), kind="stable").tolist():
        if ex.label_first[li] < 0:
            break
        cat_first.setdefault(label_to_category[labels[li]], len(cat_first))
    categories = cat_first                                          # name -> id
    n_cat = len(categories)
    cat_of_label = np.asarray([categories.get(label_to_category[lab], -1) for lab in labels], np.int32) if labels else np.zeros(0, np.int32)
    cat_arr = cat_of_label[ex.label_code] if ex.n_records else np.zeros(0, np.int32)
    sizes = np.zeros(n_cat, np.int64)
    if n_cat:
        np.add.at(sizes, cat_of_label[cat_of_label >= 0], ex.label_count[cat_of_label >= 0])
    cat_off = np.zeros(n_cat + 1, np.int64)
    np.cumsum(sizes, out=cat_off[1:])
    cuts = [split_cut_sizes(int(s), train_ratio, val_ratio, test_ratio) for s in sizes]
    n_train = np.asarray([c[0] for c in cuts], np.int64)
    n_val = np.asarray([c[1] for c in cuts], np.int64)

    # ---- device stage: rank in category -> shuffled position -> split id --------------------
    if not len(cat_arr):
        split, pos = np.zeros(0, np.uint8), np.zeros(0, np.int64)
    elif hasattr(be, "split_ids_seeded"):
        # K8 + K6: the permutations (same seed per category, :800) are made on the device and never leave it
        split, pos = be.split_ids_seeded(cat_arr, random_seed, sizes, n_train, n_val)
    else:
        perms = [be.mt19937_permutation(random_seed, int(s)) for s in sizes]
        split, pos = be.split_ids(cat_arr, np.concatenate(perms) if perms else np.zeros(0, np.int64), cat_off,
                                  n_train, n_val)
    t2 = _time.perf_counter()

    # ---- emit: ONE row-ordered walk per column scatters every record to its (category, shuffled position) slot; a category's
    # frame is a slice of those columns and its three sheets are slices of the frame, cut where the device's split id changes ----
    n_rec = ex.n_records
    slot = _pycells.category_slots(cat_arr, pos, cat_off)
    per_split = np.bincount(cat_arr.astype(np.int64) * 3 + split, minlength=3 * n_cat).reshape(n_cat, 3) if n_cat else np.zeros((0, 3), np.int64)
    fine = {"slots_s": _time.perf_counter() - t2}
    ta = _time.perf_counter()
    arrow = text_dtype == "arrow"
    text = ex.json_at_slots(slot, arrow=arrow) if n_rec else None
    fine["text_s"] = _time.perf_counter() - ta
    ta = _time.perf_counter()
    lab_arr = np.empty(len(labels), object)
    lab_arr[:] = labels
    cat_arr_names = np.empty(n_cat, object)
    cat_arr_names[:] = list(categories)
    extra = ["分类标签", "分类类别", "原始标签组合"]
    src_at_slot = []

    def idx_at_slot():
        if not src_at_slot:
            src_at_slot.append(_pycells.take(ex.src_row, None, checked=True, slot=slot))
        return src_at_slot[0]

    columns = {}
    if n_rec:
        for c in cols:
            if c not in extra and c not in present_json:
                columns[c] = _take_column(df, c, ex.src_row, slot, idx_at_slot)
        columns["分类标签"] = _pycells.take_small(lab_arr, ex.label_code, None, slot=slot, checked=True)
        columns["分类类别"] = _pycells.take_small(cat_arr_names, cat_arr, None, slot=slot, checked=True)
        columns["原始标签组合"] = _pycells.take(ex.combo_of_row, ex.src_row, checked=True, slot=slot)
    fine["columns_s"] = _time.perf_counter() - ta
    ta = _time.perf_counter()
    out_cats, cat_counts = {}, {}
    for category, cid in categories.items():
        lo, hi = int(cat_off[cid]), int(cat_off[cid + 1])
        if arrow:
            import pyarrow as pa

            tbuf, toff = text
            piece = pd.arrays.ArrowStringArray(pa.chunked_array([pa.LargeStringArray.from_buffers(
                hi - lo, pa.py_buffer(toff[lo:hi + 1]), pa.py_buffer(tbuf))]))
        else:
            piece = text[lo:hi]
        data = {}
        for c in cols:                                               # a column named like a new one is overwritten in place (:769-771)
            data[c] = piece if (c in present_json and c not in extra) else columns[c][lo:hi]
        for c in extra:
            data[c] = columns[c][lo:hi]
        frame = pd.DataFrame(data, copy=False)
        a, b = int(per_split[cid, 0]), int(per_split[cid, 0] + per_split[cid, 1])
        out_cats[category] = (frame.iloc[:a], frame.iloc[a:b], frame.iloc[b:])
        cat_counts[category] = hi - lo
    fine["frames_s"] = _time.perf_counter() - ta
    t3 = _time.perf_counter()

    if n:
        sources = _column_values(df, "source") if "source" in df.columns else np.full(n, None, object)
        counts = pd.DataFrame({"source": sources, "原始标签组合": ex.combo_of_row, "拆分条数": ex.n_out,
                               "是否可分类": _VERDICTS[ex.verdict_code], "无法分类原因": ex.reasons_of_row}, copy=False)
    else:
        counts = pd.DataFrame()
    if len(ex.unc_row):                   # rows in the reference's append order (:724, :748, :756, :780)
        data = {c: _take_column(df, c, ex.unc_row) for c in cols if c != "无法分类原因" and c != "无法分类标签"}
        tail = {"无法分类原因": ex.unc_reason}
        has_label = ex.unc_has_label
        if has_label.any():
            tail["无法分类标签"] = np.where(has_label, ex.unc_label, np.nan)
        elif "无法分类标签" in cols:
            tail["无法分类标签"] = _take_column(df, "无法分类标签", ex.unc_row)
        for c in cols:
            if c in tail:
                data[c] = tail.pop(c)
        data.update(tail)
        unc = pd.DataFrame(data, copy=False)
        unc.index = df.index.take(ex.unc_row)                      # DataFrame(list of row copies) keeps the rows' labels
    else:
        unc = pd.DataFrame()
    result = {"categories": out_cats, "unclassified": unc, "split_counts": counts,
              "category_counts": cat_counts,
              "expanded": {"src_row": ex.src_row, "category_id": cat_arr, "position": pos, "split": split,
                           "category_names": list(categories)}}
    ex.close()
    del keep_alive
    if stats is not None:
        t4 = _time.perf_counter()
        stats.update({"expand_s": t1 - t0, "device_s": t2 - t1, "category_frames_s": t3 - t2, "side_tables_s": t4 - t3,
                      "records": int(n_rec), "fast_cells": int(ex.native.fast_cells) if ex.native is not None else 0})
        stats["category_frames_fine"] = {k: round(v, 4) for k, v in fine.items()}
        if ex.native is not None:
            stats["native_s"] = dict(ex.native.seconds)
            stats["expand_batches"] = len(getattr(ex.native, "_parts", (None,)))
    return result


def split_dataset_by_rules(
        input_csv_path: str,
        rules_excel_path: str,
        output_dir: str,
        rule_mode: str = "wide",
        sheet_name: Optional[str] = None,
        label_col: Optional[str] = None,
        category_col: Optional[str] = None,
        json_columns: Optional[list] = None,
        train_ratio: float = 0.8,
        val_ratio: float = 0.1,
        test_ratio: float = 0.1,
        random_seed: int = 42,
        backend=None,
):
    if not os.path.exists(input_csv_path):
        raise FileNotFoundError(f"输入CSV不存在：{input_csv_path}")
    if not os.path.exists(rules_excel_path):
        raise FileNotFoundError(f"规则Excel不存在：{rules_excel_path}")

    df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
    rules_df = pd.read_excel(rules_excel_path, sheet_name=sheet_name) if sheet_name else pd.read_excel(rules_excel_path)
    label_to_category = rules_to_label_map(rules_df, rule_mode, label_col, category_col)

    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    res = split_frames(df, label_to_category, json_columns, train_ratio, val_ratio, test_ratio, random_seed, backend)

    category_files = []
    for category, (train_df, val_df, test_df) in res["categories"].items():
        out_path = output_dir / f"{safe_filename(category)}.xlsx"
        with pd.ExcelWriter(out_path) as writer:
            train_df.to_excel(writer, sheet_name="train", index=False)
            val_df.to_excel(writer, sheet_name="val", index=False)
            test_df.to_excel(writer, sheet_name="test", index=False)
        category_files.append(out_path)
    unclassified_path = output_dir / "unclassified.xlsx"
    res["unclassified"].to_excel(unclassified_path, index=False)
    split_counts_path = output_dir / "split_counts.xlsx"
    res["split_counts"].to_excel(split_counts_path, index=False)

    return {
        "output_dir": output_dir,
        "category_files": category_files,
        "unclassified": unclassified_path,
        "split_counts": split_counts_path,
        "summary": {
            "categories": len(res["categories"]),
            "classified": sum(res["category_counts"].values()),
            "unclassified": len(res["unclassified"]),
            "category_counts": res["category_counts"],
        },
    }


# =============================================================================== f4  YOLO label lines
REASON_NO_MATCHING_BOX = "无匹配标签框"          # reference processor.py:1009
REASON_NO_IMAGE_SIZE = "缺少图像尺寸"            # :1024
REASON_NO_VALID_BOX = "标注框无效"               # :1058
_EXACT_INT = 1 << 52                             # |x1 + x2| stays exact in f64 below 2^53


def _label_lines_python(boxes, class_id, width, height) -> list:
    """The reference's own arithmetic on Python numbers (processor.py:1046-1052) for the rows the device does
    not print: big integers (exact int / int division), bools, values of 2^43 and more."""
    out = []
    for _, xa, ya, xb, yb in boxes:
        left, right = min(xa, xb), max(xa, xb)
        top, bottom = min(ya, yb), max(ya, yb)
        box_w = max(right - left, 0.0)
        box_h = max(bottom - top, 0.0)
        if box_w <= 0 or box_h <= 0:
            continue
        out.append(f"{class_id} {(left + right) / 2 / width:.6f} {(top + bottom) / 2 / height:.6f} "
                   f"{box_w / width:.6f} {box_h / height:.6f}")
    return out


def _plain_number(v, limit=_EXACT_INT) -> bool:
    t = type(v)
    if t is float or t is np.float64:
        return True
    if t is int or (isinstance(v, np.integer)):
        return -limit <= int(v) <= limit
    return False


def _yolo_rows_python(rows, cells, label_values, class_ids, widths, heights, be, texts, reasons, stats):
    """rows of the batch through CPython's json (box extraction) + K7; everything irregular on Python numbers"""
    dev_rows, dev_boxes, row_off, dev_w, dev_h, dev_cid = [], [], [0], [], [], []
    py_rows = {}
    for i in rows:
        boxes = [b for b in _extract_boxes_with_labels(cells[i]) if b[0] == label_values[i]]
        if not boxes:
            reasons[i] = REASON_NO_MATCHING_BOX
            continue
        w, h = widths[i], heights[i]
        if not w or not h:
            reasons[i] = REASON_NO_IMAGE_SIZE
            continue
        cid = class_ids[i]
        if (_plain_number(w, 1 << 53) and _plain_number(h, 1 << 53) and isinstance(cid, (int, np.integer)) and 0 <= cid < (1 << 31)
                and all(_plain_number(v) for b in boxes for v in b[1:])):
            dev_rows.append(i)
            for b in boxes:
                dev_boxes.extend(b[1:])
            row_off.append(len(dev_boxes) // 4)
            dev_w.append(w)
            dev_h.append(h)
            dev_cid.append(cid)
        else:
            py_rows[i] = boxes
    if dev_rows:
        off, flag, data = be.yolo_lines(np.asarray(dev_boxes, np.float64), np.asarray(row_off, np.int32), None,
                                        np.asarray(dev_w, np.float64), np.asarray(dev_h, np.float64),
                                        np.asarray(dev_cid, np.int32))
        for k, i in enumerate(dev_rows):
            if flag[k] == 0:
                texts[i] = data[off[k]:off[k + 1]].decode("ascii")
            elif flag[k] == 1:
                reasons[i] = REASON_NO_VALID_BOX
            else:                                            # the device leaves huge values to the host
                b0, b1 = row_off[k], row_off[k + 1]
                py_rows[i] = [(label_values[i], *dev_boxes[4 * b:4 * b + 4]) for b in range(b0, b1)]
    for i, boxes in py_rows.items():
        lines = _label_lines_python(boxes, class_ids[i], widths[i], heights[i])
        if lines:
            texts[i] = "\n".join(lines)
        else:
            reasons[i] = REASON_NO_VALID_BOX
    stats["device_rows"] += len(dev_rows)
    stats["python_rows"] += len(py_rows)


def _numeric_sizes(values):
    """-> float64 array when every value is a plain int / float below 2^53 (so `not v` is `v == 0`), else None"""
    if pd.api.types.infer_dtype(values, skipna=False) not in ("integer", "floating", "mixed-integer-float"):
        return None
    arr = np.asarray(values, np.float64)
    return arr if not (np.abs(arr[np.isfinite(arr)]) > float(1 << 53)).any() else None


def yolo_label_texts(cells, label_values, class_ids, widths, heights, backend=None, stats: Optional[dict] = None):
    """Label-file text per row of a split sheet: the part of generate_yolo_datasets_from_excels between the
    box extraction and ``label_path.write_text`` (reference processor.py:1004-1060), batched.

    cells[i] is the row's annotation JSON, label_values[i] = str(row[label_col]) (:992), class_ids[i] =
    class_to_id[label_value] (:1049), widths / heights the row's image size (:1013-1014).
    -> (texts, reasons): texts[i] is "\n".join(label_lines) or None; reasons[i] is the reference's skip
    reason for a None (无匹配标签框 / 缺少图像尺寸 / 标注框无效) in the order the reference tests them.
    Host: native labelled-box scan (csrc/host_json.cpp; CPython json for irregular cells) + label match;
    device: K7 (arithmetic, exact "%.6f", joining)."""
    be = _backend(backend)
    n = len(cells)
    texts, reasons = [None] * n, [None] * n
    st = {"rows": n, "device_rows": 0, "python_rows": 0, "python_cells": 0}
    rest = range(n)
    w_arr, h_arr = _numeric_sizes(widths), _numeric_sizes(heights)
    cid_ok = n > 0 and pd.api.types.infer_dtype(class_ids, skipna=False) == "integer"
    if n and _nj.enabled() and w_arr is not None and h_arr is not None and cid_ok and all(type(v) is str for v in label_values):
        cid_arr = np.asarray(class_ids, np.int64)
        try:
            scan = _nj.scan_labelled(cells, label_values) if ((cid_arr >= 0) & (cid_arr < (1 << 31))).all() else None
        except UnicodeEncodeError:
            scan = None
        if scan is not None:
            regular = scan.status != _nj.IRREGULAR
            counts = np.diff(scan.cell_box_off)
            n_sel = np.add.reduceat(np.concatenate([scan.sel, [0]]).astype(np.int64), scan.cell_box_off[:-1].astype(np.int64)) \
                if scan.n_boxes else np.zeros(n, np.int64)
            n_sel = np.where(counts > 0, n_sel, 0)
            no_box = regular & (n_sel == 0)
            no_size = regular & ~no_box & ((w_arr == 0) | (h_arr == 0))
            dev = np.flatnonzero(regular & ~no_box & ~no_size)
            for i in np.flatnonzero(no_box).tolist():
                reasons[i] = REASON_NO_MATCHING_BOX
            for i in np.flatnonzero(no_size).tolist():
                reasons[i] = REASON_NO_IMAGE_SIZE
            if len(dev):
                # the batch keeps every scanned box: rows outside `dev` get a zero image size, which K7 flags and skips
                w_dev, h_dev = np.zeros(n), np.zeros(n)
                w_dev[dev], h_dev[dev] = w_arr[dev], h_arr[dev]
                off, flag, data = be.yolo_lines(scan.box4, scan.cell_box_off, scan.sel, w_dev, h_dev, cid_arr.astype(np.int32))
                import pyarrow as pa
                strs = pa.LargeStringArray.from_buffers(n, pa.py_buffer(np.ascontiguousarray(off)),
                                                        pa.py_buffer(data if data else b"\0")).to_numpy(zero_copy_only=False)
                host_rows = []
                for i in dev.tolist():
                    f = flag[i]
                    if f == 0:
                        texts[i] = strs[i]
                    elif f == 1:
                        reasons[i] = REASON_NO_VALID_BOX
                    else:
                        host_rows.append(i)
                st["device_rows"] += len(dev) - len(host_rows)
                for i in host_rows:                             # values of 2^43 and more: printed from the scanned f64 boxes
                    b0, b1 = int(scan.cell_box_off[i]), int(scan.cell_box_off[i + 1])
                    boxes = [(label_values[i], *scan.box4[b].tolist()) for b in range(b0, b1) if scan.sel[b]]   # ints <= 2^52 print the same as floats
                    lines = _label_lines_python(boxes, class_ids[i], widths[i], heights[i])
                    texts[i], reasons[i] = ("\n".join(lines), None) if lines else (None, REASON_NO_VALID_BOX)
                    st["python_rows"] += 1
            rest = np.flatnonzero(~regular).tolist()
            scan.close()
    st["python_cells"] = len(rest)
    if len(rest):
        _yolo_rows_python(rest, cells, label_values, class_ids, widths, heights, be, texts, reasons, st)
    if stats is not None:
        stats.update(st)
    return texts, reasons


def generate_yolo_datasets_from_excels(
        category_excels: list,
        output_dir: str,
        image_cache_dir: Optional[str] = None,
        source_col: str = "source",
        label_col: str = "分类标签",
        json_col_primary: str = BBOX_COL,
        json_col_fallback: str = ANNOTATION_COL,
        width_col: str = "width",
        height_col: str = "height",
        download_images: bool = True,
        random_seed: int = 42,
        class_order: Optional[list] = None,
        resume: bool = True,
        progress_callback=None,
        backend=None,
):
    """Drop-in for reference processor.py:893-1093: one YOLO dataset directory per category workbook
    (images/<split>, labels/<split>, data.yaml) plus yolo_skipped.xlsx.

    Per split sheet the rows are shuffled like ``sample(frac=1, random_state=seed)`` (host MT19937), the label
    texts of all rows are produced in one batch (``yolo_label_texts`` -> K7) and the per-row side effects (resume
    check, image copy, label file, skip records) are then replayed in the reference's order.  Rows whose image
    size comes from the image file rather than from the sheet (:1015-1020) are printed on the host."""
    import yaml

    be = _backend(backend)
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    cache_dir = Path(image_cache_dir) if image_cache_dir else (output_dir / "image_cache")
    cache_dir.mkdir(parents=True, exist_ok=True)

    datasets, dataset_name_map, skipped, dataset_stats = [], {}, [], {}
    total_rows = processed_rows = downloaded_images = 0
    used_dir_names = set()
    splits = ["train", "val", "test"]

    for excel_path in category_excels:                                   # :917-924
        if not excel_path or not Path(excel_path).exists():
            continue
        book = pd.ExcelFile(excel_path)
        for split in splits:
            if split in book.sheet_names:
                total_rows += len(pd.read_excel(excel_path, sheet_name=split))

    last = None                                                          # arguments of the closing progress call
    for idx_excel, excel_path in enumerate(category_excels):
        if not excel_path or not Path(excel_path).exists():
            continue
        excel_path = Path(excel_path)
        category_name = excel_path.stem
        base_dir_name = safe_filename(str(category_name)) if category_name else f"category_{idx_excel:03d}"
        dir_name, suffix = base_dir_name, 1
        while dir_name in used_dir_names:                                # :931-936
            dir_name = f"{base_dir_name}_{suffix}"
            suffix += 1
        used_dir_names.add(dir_name)
        dataset_dir = output_dir / dir_name
        dataset_name_map[dataset_dir.name] = category_name
        images_root, labels_root = dataset_dir / "images", dataset_dir / "labels"
        for split in splits:
            (images_root / split).mkdir(parents=True, exist_ok=True)
            (labels_root / split).mkdir(parents=True, exist_ok=True)

        book = pd.ExcelFile(excel_path)
        split_sheets = [sp for sp in splits if sp in book.sheet_names]
        all_labels, frames = [], {}
        for split in split_sheets:
            frames[split] = pd.read_excel(excel_path, sheet_name=split)
            if label_col in frames[split].columns:
                all_labels.extend(str(v) for v in frames[split][label_col].dropna())
        classes = sorted(dict.fromkeys(all_labels))                      # :958-963
        if class_order:
            head = [c for c in class_order if c in classes]
            classes = head + [c for c in classes if c not in head]
        class_to_id = {name: i for i, name in enumerate(classes)}
        dataset_stats[category_name] = {sp: 0 for sp in splits}

        for split in split_sheets:
            frame = frames[split]
            order = be.mt19937_permutation(random_seed, len(frame))      # = sample(frac=1, random_state=seed) (:969)
            frame = frame.iloc[order].reset_index(drop=True)
            columns = set(frame.columns)
            get = lambda name, default=None: (frame[name].tolist() if name in columns else [default] * len(frame))  # noqa: E731
            sources, widths, heights = get(source_col), get(width_col), get(height_col)
            labels = [str(v) for v in get(label_col, "")]
            primary, fallback = get(json_col_primary), get(json_col_fallback)
            cells = [a or b for a, b in zip(primary, fallback)]          # row.get(primary) or row.get(fallback) (:1004)
            usable = [bool(src) and bool(lab) and lab in class_to_id for src, lab in zip(sources, labels)]
            batch = [i for i, ok in enumerate(usable) if ok]
            texts, reasons = yolo_label_texts([cells[i] for i in batch], [labels[i] for i in batch],
                                              [class_to_id[labels[i]] for i in batch], [widths[i] for i in batch],
                                              [heights[i] for i in batch], be)
            text_of = dict(zip(batch, zip(texts, reasons)))

            for idx in range(len(frame)):
                last = (processed_rows, total_rows, downloaded_images, category_name, split, f"idx_{idx}", "", excel_path.name, idx)
                if progress_callback and processed_rows % 50 == 0:
                    progress_callback(*last)
                processed_rows += 1                                      # every path below counts the row once
                source, label_value = sources[idx], labels[idx]
                if not source:
                    skipped.append({"category": category_name, "reason": "缺少source", "split": split})
                    continue
                if not label_value or label_value not in class_to_id:
                    skipped.append({"category": category_name, "reason": "缺少或无效分类标签", "split": split})
                    continue
                label_path = labels_root / split / f"{_safe_image_stem(str(source), idx)}.txt"
                if resume and label_path.exists() and label_path.stat().st_size > 0:
                    dataset_stats[category_name][split] += 1
                    continue
                text, reason = text_of[idx]
                if reason == REASON_NO_MATCHING_BOX:
                    skipped.append({"category": category_name, "reason": reason, "split": split})
                    continue
                image_path = None
                if download_images:
                    image_path = _ensure_image_cached(str(source), cache_dir)
                elif Path(str(source)).exists():
                    image_path = Path(str(source))
                width, height = widths[idx], heights[idx]
                if (not width or not height) and image_path:             # size from the image file (:1015-1020)
                    try:
                        from PIL import Image
                        with Image.open(image_path) as img:
                            width, height = img.size
                        boxes = [b for b in _extract_boxes_with_labels(cells[idx]) if b[0] == label_value]
                        lines = _label_lines_python(boxes, class_to_id[label_value], width, height)
                        text, reason = ("\n".join(lines), None) if lines else (None, REASON_NO_VALID_BOX)
                    except Exception:  # noqa: BLE001
                        pass
                if not width or not height:
                    skipped.append({"category": category_name, "reason": REASON_NO_IMAGE_SIZE, "split": split})
                    continue
                if not image_path:
                    skipped.append({"category": category_name, "reason": "图片下载失败", "split": split})
                    continue
                out_image = images_root / split / f"{label_path.stem}{image_path.suffix}"
                if not out_image.exists():
                    try:
                        out_image.write_bytes(Path(image_path).read_bytes())
                        downloaded_images += 1
                    except Exception:  # noqa: BLE001
                        skipped.append({"category": category_name, "reason": "图片写入失败", "split": split})
                        continue
                if text is not None:
                    label_path.write_text(text, encoding="utf-8")
                    dataset_stats[category_name][split] += 1
                else:
                    skipped.append({"category": category_name, "reason": REASON_NO_VALID_BOX, "split": split})

        (dataset_dir / "data.yaml").write_text(yaml.dump({
            "path": str(dataset_dir), "train": "images/train", "val": "images/val", "test": "images/test",
            "nc": len(classes), "names": classes}, sort_keys=False, allow_unicode=True), encoding="utf-8")
        datasets.append(dataset_dir)

    skipped_path = output_dir / "yolo_skipped.xlsx"
    pd.DataFrame(skipped if skipped else [{"category": "无", "reason": "无", "split": "无"}]).to_excel(skipped_path, index=False)
    if progress_callback and last is not None:
        # the reference's closing call reads names it never defines (:1076-1077, NameError); report the last row instead
        progress_callback(processed_rows, *last[1:])
    return {"datasets": datasets, "skipped": skipped_path, "stats": dataset_stats, "total": total_rows,
            "processed": processed_rows, "downloaded": downloaded_images, "dataset_name_map": dataset_name_map}


# =============================================================================== label_replace (pipeline step between a4 and a5)
def mapping_to_label_map(mapping_df: pd.DataFrame, old_col: Optional[str] = None, new_col: Optional[str] = None) -> dict:
    """old label -> new label from the mapping sheet (reference processor.py:532-545): the first two columns unless
    named; a pair with a blank or a "nan" spelling on either side is dropped; later rows overwrite earlier ones."""
    if not old_col or not new_col:
        cols = list(mapping_df.columns)
        if len(cols) < 2:
            raise ValueError("标签对照表至少需要两列")
        old_col, new_col = old_col or cols[0], new_col or cols[1]

    names = list(mapping_df.columns)
    values = mapping_df.values            # what iterrows walks: one common dtype for the row (ints next to floats print as floats)

    def texts(col):   # str(row.get(col, "")).strip() for every row
        if col not in names:
            return [""] * len(mapping_df)
        return [str(v).strip() for v in values[:, names.index(col)]]

    def usable(t):
        return bool(t) and t.lower() != "nan"

    return {o: n for o, n in zip(texts(old_col), texts(new_col)) if usable(o) and usable(n)}


def _relabel_name(raw_name, label_map):
    """reference utils.py:664-679: (new name, labels replaced, labels seen); the new name is the SET of the
    replaced labels, sorted, joined with ',' — also when nothing was replaced"""
    if not raw_name:
        return raw_name, 0, 0
    tokens = _split_object_labels(raw_name)
    hits = sum(1 for t in tokens if t in label_map)
    return ",".join(sorted({label_map.get(t, t) for t in tokens})), hits, len(tokens)


class _RelabelTotals:
    __slots__ = ("total_objects", "total_labels", "replaced_labels", "replaced_objects", "invalid_json_rows",
                 "missing_name_objects", "unmatched")

    def __init__(self):
        self.total_objects = self.total_labels = self.replaced_labels = self.replaced_objects = 0
        self.invalid_json_rows = self.missing_name_objects = 0
        self.unmatched = {}            # label -> occurrences, in first-seen order (the unmatched sheet keeps it among ties)


def _relabel_cell(cell: str, label_map: dict, tot: _RelabelTotals):
    """One annotation cell (reference processor.py:572-609) -> (rewritten text or None when the cell stays as it is,
    [(old name, new name)] of the objects whose name would change, any object renamed?).  Every cell whose document
    has a list under "objects" is re-serialised (``json.dumps(..., ensure_ascii=False)``), renamed or not; a document
    that is not an object, or a name that is neither text nor empty, ends the step with the reference's exception."""
    try:
        doc = json.loads(cell)
    except json.JSONDecodeError:
        tot.invalid_json_rows += 1
        return None, (), False
    objects = doc.get("objects")
    if not isinstance(objects, list):
        return None, (), False
    changes, renamed = [], False
    unmatched = tot.unmatched
    for obj in objects:
        if not isinstance(obj, dict):
            continue
        tot.total_objects += 1
        raw = obj.get("name")
        if raw is None:
            tot.missing_name_objects += 1
            continue
        for lbl in _split_object_labels(raw):
            if lbl not in label_map:
                unmatched[lbl] = unmatched.get(lbl, 0) + 1
        new_name, hits, seen = _relabel_name(raw, label_map)
        tot.total_labels += seen
        if hits:
            obj["name"] = new_name
            tot.replaced_labels += hits
            tot.replaced_objects += 1
            renamed = True
        if raw != new_name:
            changes.append((raw, new_name))
    return json.dumps(doc, ensure_ascii=False), changes, renamed


def _relabel_cells_python(cells, label_map, tot: _RelabelTotals):
    """every cell through CPython json, in order -> (new text or None, joined old names or None, joined new names, renamed?) per cell"""
    out = []
    for cell in cells:
        if not isinstance(cell, str) or not cell:          # NaN, numbers, ""
            out.append((None, None, None, False))
            continue
        text, changes, renamed = _relabel_cell(cell, label_map, tot)
        if changes:
            out.append((text, "；".join([a for a, _ in changes]), "；".join([b for _, b in changes]), renamed))
        else:
            out.append((text, None, None, renamed))
    return out


def _relabel_add_counts(tot: _RelabelTotals, r):
    done = r.status == _nj.RL_REWRITTEN
    sums = r.counts[done].sum(axis=0, dtype=np.int64) if done.any() else np.zeros(5, np.int64)
    tot.total_objects += int(sums[0]); tot.missing_name_objects += int(sums[1]); tot.total_labels += int(sums[2])
    tot.replaced_labels += int(sums[3]); tot.replaced_objects += int(sums[4])
    tot.invalid_json_rows += int(np.count_nonzero(r.status == _nj.RL_UNDECODABLE))


def _relabel_add_unmatched(tot: _RelabelTotals, token):
    """labels in order of appearance -> counts in first-seen order"""
    if len(token):
        codes, uniques = pd.factorize(token)               # uniques in order of first appearance
        counts = np.bincount(codes, minlength=len(uniques))
        for lbl, cnt in zip(uniques.tolist(), counts.tolist()):
            tot.unmatched[lbl] = tot.unmatched.get(lbl, 0) + cnt


def _relabel_cells_native(cells, label_map, tot: _RelabelTotals):
    """the same through the native relabeller (csrc/host_json.cpp); the cells it calls irregular go through CPython
    in their turn, so counters, first-seen order of the unmatched labels and the first exception are the reference's"""
    try:
        r = _nj.relabel(cells, label_map)
    except UnicodeEncodeError:                             # a lone surrogate somewhere: CPython path for the batch
        return _relabel_cells_python(cells, label_map, tot)
    done = r.status == _nj.RL_REWRITTEN
    irregular = np.flatnonzero(r.status == _nj.RL_IRREGULAR)
    _relabel_add_counts(tot, r)
    texts = np.where(done, r.text(), None)
    before = np.where(r.has_diff != 0, r.before(), None)
    after = np.where(r.has_diff != 0, r.after(), None)
    renamed = r.counts[:, 4] > 0
    r.close()
    out = list(zip(texts.tolist(), before.tolist(), after.tolist(), renamed.tolist()))
    token, token_cell = r.token, r.token_cell
    if len(irregular):
        extra_tok, extra_cell = [], []
        for i in irregular.tolist():
            side = _RelabelTotals()
            res = _relabel_cells_python([cells[i]], label_map, side)[0]          # may raise, like the reference
            out[i] = res
            for k in ("total_objects", "total_labels", "replaced_labels", "replaced_objects", "invalid_json_rows", "missing_name_objects"):
                setattr(tot, k, getattr(tot, k) + getattr(side, k))
            for lbl, cnt in side.unmatched.items():      # first-seen order within the cell; the counts are merged below
                extra_tok += [lbl] * cnt
                extra_cell += [i] * cnt
        if extra_tok:
            token = np.concatenate([token, np.array(extra_tok, dtype=object)])
            token_cell = np.concatenate([token_cell, np.array(extra_cell, dtype=np.int64)])
            order = np.argsort(token_cell, kind="stable")
            token = token[order]
    _relabel_add_unmatched(tot, token)
    return out


def replace_labels_frame(df: pd.DataFrame, label_map: dict, json_columns: Optional[list] = None):
    """DataFrame twin of replace_labels_by_mapping: (frame with the rewritten cells, summary counters, diff rows,
    unmatched label counts).  Cells are visited row by row, and within a row in ``json_columns`` order — the order
    of the diff rows and of first-seen unmatched labels."""
    if json_columns is None:
        json_columns = [c for c in (BBOX_COL, ANNOTATION_COL) if c in df.columns]
    present = [c for c in json_columns if c in df.columns]
    tot = _RelabelTotals()
    n, k = len(df), len(present)
    columns = [df[c].tolist() for c in present]
    cells = [None] * (n * k)                               # row-major: the order the reference walks them in
    for j, col in enumerate(columns):
        cells[j::k] = col
    relabel = _relabel_cells_native if _nj.enabled() else _relabel_cells_python
    res = []
    for start in range(0, len(cells), _NATIVE_CHUNK_CELLS):
        res += relabel(cells[start:start + _NATIVE_CHUNK_CELLS], label_map, tot)
    sources = df["source"].tolist() if "source" in df.columns else None
    diff_rows = []
    row_renamed = np.zeros(n, bool)
    for idx, (text, before, after, renamed) in enumerate(res):
        if text is None:
            continue
        i, j = divmod(idx, k)
        columns[j][i] = text
        if renamed:
            row_renamed[i] = True
        if before is not None:
            diff_rows.append({"source": sources[i] if sources is not None else None, "column": present[j], "before": before, "after": after})
    out = df.copy()
    for j, c in enumerate(present):
        if out[c].dtype != object:
            continue                                           # a numeric column holds no cell to rewrite
        out[c] = pd.Series(columns[j], index=out.index, dtype=object)
    counters = {"replaced_rows": int(row_renamed.sum()), "total_objects": tot.total_objects, "replaced_objects": tot.replaced_objects,
                "total_labels": tot.total_labels, "replaced_labels": tot.replaced_labels,
                "invalid_json_rows": tot.invalid_json_rows, "missing_name_objects": tot.missing_name_objects}
    return out, counters, diff_rows, tot.unmatched


def _relabel_csv_fast(input_csv_path, label_map, output_csv_path, json_columns):
    """CSV -> CSV label replacement without pandas touching the JSON columns (fastcsv + native relabeller).
    Returns NotImplemented when the fast path does not apply (nothing has been written then), else
    (n_rows, counters, diff_rows, unmatched)."""
    try:
        table = _fc.read_split(str(input_csv_path), [BBOX_COL, ANNOTATION_COL] if json_columns is None else list(json_columns))
    except (OSError, ValueError, pd.errors.ParserError, UnicodeDecodeError):
        return NotImplemented
    if table is None:
        return NotImplemented
    if json_columns is None:
        json_columns = [c for c in (BBOX_COL, ANNOTATION_COL) if c in table.names]
    present = [c for c in json_columns if c in table.names]
    if not present or len(set(present)) != len(present) or any(c not in table.heavy for c in present):
        return NotImplemented                              # short or numeric JSON columns: pandas types them, pandas reads them
    label_map = label_map()
    k = len(present)
    tot = _RelabelTotals()
    new_cols, tokens, token_keys, diffs, runs = {}, [], [], [], []
    row_renamed = np.zeros(table.n_rows, bool)
    for j, c in enumerate(present):
        col = table.heavy[c]
        r = _nj.relabel_buffers(col.data, col.off, col.na, label_map, keep=col)
        runs.append(r)
        _relabel_add_counts(tot, r)
        row_renamed |= r.counts[:, 4] > 0
        tokens.append(r.token)
        token_keys.append(r.token_cell * k + j)
        diff_cells = np.flatnonzero(r.has_diff)
        if len(diff_cells):
            before, after = r.before(), r.after()
            diffs += [(int(i) * k + j, before[i], after[i]) for i in diff_cells.tolist()]
        irregular = np.flatnonzero(r.status == _nj.RL_IRREGULAR)
        spliced = None
        for i in irregular.tolist():                        # CPython decides, in its turn; may raise, like the reference
            side = _RelabelTotals()
            text, before, after, renamed = _relabel_cells_python([col.cell(i)], label_map, side)[0]
            for name in ("total_objects", "total_labels", "replaced_labels", "replaced_objects", "invalid_json_rows", "missing_name_objects"):
                setattr(tot, name, getattr(tot, name) + getattr(side, name))
            for lbl, cnt in side.unmatched.items():
                tokens.append(np.array([lbl] * cnt, dtype=object))
                token_keys.append(np.full(cnt, i * k + j, np.int64))
            row_renamed[i] |= renamed
            if before is not None:
                diffs.append((i * k + j, before, after))
            if text is not None:
                if spliced is None:
                    spliced = r.text().tolist()
                spliced[i] = text
        if spliced is None:
            data, off = r.text_buffers()
            new_cols[c] = _fc.Utf8Column(data, off, col.na, r)
        else:
            for i in np.flatnonzero(col.na).tolist():
                spliced[i] = None
            spec = _fc._series_column(pd.Series(spliced, dtype=object))
            new_cols[c] = _fc.Utf8Column(spec[1], spec[2], spec[3])
    if tokens:
        token = np.concatenate(tokens)
        key = np.concatenate(token_keys)
        _relabel_add_unmatched(tot, token[np.argsort(key, kind="stable")])
    diffs.sort(key=lambda d: d[0])
    sources = table.light["source"].tolist() if "source" in table.light.columns else None
    diff_rows = [{"source": sources[key // k] if sources is not None else None, "column": present[key % k], "before": b, "after": a}
                 for key, b, a in diffs]
    columns = [new_cols[nm] if nm in new_cols else (table.heavy[nm] if nm in table.heavy else table.light[nm]) for nm in table.names]
    Path(output_csv_path).parent.mkdir(parents=True, exist_ok=True)
    ok = _fc.write_table(str(output_csv_path), table.names, columns, table.n_rows)
    for r in runs:
        r.close()
    if not ok:
        return NotImplemented
    counters = {"replaced_rows": int(row_renamed.sum()), "total_objects": tot.total_objects, "replaced_objects": tot.replaced_objects,
                "total_labels": tot.total_labels, "replaced_labels": tot.replaced_labels,
                "invalid_json_rows": tot.invalid_json_rows, "missing_name_objects": tot.missing_name_objects}
    return table.n_rows, counters, diff_rows, tot.unmatched


def replace_labels_by_mapping(
        input_csv_path: str,
        mapping_excel_path: str,
        output_csv_path: str,
        sheet_name: Optional[str] = None,
        old_col: Optional[str] = None,
        new_col: Optional[str] = None,
        json_columns: Optional[list] = None,
        diff_excel_path: Optional[str] = None,
        unmatched_excel_path: Optional[str] = None,
        sample_size: int = 30,
):
    """Drop-in for reference processor.py:516-652 (pipeline step ``label_replace``, between the IoU filter and the
    split): object names rewritten through the mapping sheet, every parsed cell re-serialised, a diff sheet and an
    unmatched-label sheet on request.  Host-only step: there is no arithmetic for the device in it."""
    output_csv_path = Path(output_csv_path)
    made = {}

    def label_map():                                       # the mapping sheet is read after the CSV, as in the reference
        if "map" not in made:
            mapping_df = pd.read_excel(mapping_excel_path, sheet_name=sheet_name) if sheet_name else pd.read_excel(mapping_excel_path)
            made["map"] = mapping_to_label_map(mapping_df, old_col, new_col)
        return made["map"]

    fast = NotImplemented
    if _fc.enabled() and _nj.enabled() and os.path.isfile(str(input_csv_path)):
        fast = _relabel_csv_fast(input_csv_path, label_map, output_csv_path, json_columns)
    if fast is not NotImplemented:
        LAST_IO_PATH["label_replace"] = "native"
        total_rows, counters, diff_rows, unmatched = fast
    else:
        LAST_IO_PATH["label_replace"] = "pandas"
        df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
        out, counters, diff_rows, unmatched = replace_labels_frame(df, label_map(), json_columns)
        total_rows = len(df)
        output_csv_path.parent.mkdir(parents=True, exist_ok=True)
        out.to_csv(output_csv_path, index=False, encoding="utf-8-sig")
    label_map = label_map()

    diff_path = None
    if diff_excel_path:
        diff_path = Path(diff_excel_path)
        diff_path.parent.mkdir(parents=True, exist_ok=True)
        pd.DataFrame(diff_rows).to_excel(diff_path, index=False)
    unmatched_path = None
    if unmatched_excel_path:
        unmatched_path = Path(unmatched_excel_path)
        unmatched_path.parent.mkdir(parents=True, exist_ok=True)
        if unmatched:
            sheet = pd.DataFrame([{"标签": k, "数量": v} for k, v in unmatched.items()]).sort_values("数量", ascending=False)
        else:
            sheet = pd.DataFrame(columns=["标签", "数量"])
        sheet.to_excel(unmatched_path, index=False)

    summary = {"total_rows": total_rows, "replaced_rows": counters["replaced_rows"], "total_objects": counters["total_objects"],
               "replaced_objects": counters["replaced_objects"], "total_labels": counters["total_labels"],
               "replaced_labels": counters["replaced_labels"], "invalid_json_rows": counters["invalid_json_rows"],
               "missing_name_objects": counters["missing_name_objects"], "mapping_size": len(label_map),
               "unmatched_labels": len(unmatched)}
    return {"output_csv": output_csv_path, "summary": summary, "diff": diff_path, "unmatched": unmatched_path,
            "sample_diff": diff_rows[:sample_size]}


def overwrite_reference_with_result(result_csv: str, ref_csv: str):
    """Drop-in for reference processor.py:221-227: the filtered result becomes the next run's reference table."""
    import shutil

    if not os.path.exists(result_csv):
        raise FileNotFoundError(f"结果文件不存在：{result_csv}")
    shutil.copy2(result_csv, ref_csv)


# =============================================================================== summaries either side of a5 / f4
_UNDEFINED_LABEL_REASON = re.compile(r"^标签(.+?)(未在规则中定义)$")          # reference processor.py:860


def unclassified_summary_frames(df: pd.DataFrame) -> dict:
    """The three sheets of unclassified_summary.xlsx (reference processor.py:852-885): rows per reason (NaN counted as
    未知原因), rows per label, rows per (label, reason).  The labels of a row come from its 无法分类标签 cell, else from a
    reason of the form 标签<label>未在规则中定义, else the row counts under 无标签.  Ties keep pandas' own order."""
    reason_col, label_col = "无法分类原因", "无法分类标签"
    n = len(df)
    reasons = df[reason_col].tolist() if reason_col in df.columns else ["未知原因"] * n
    reason_series = df[reason_col] if reason_col in df.columns else pd.Series(reasons, index=df.index, name=reason_col)
    reason_counts = reason_series.fillna("未知原因").value_counts().reset_index()
    reason_counts.columns = ["原因", "数量"]
    label_cells = df[label_col].tolist() if label_col in df.columns else [None] * n

    by_label, by_pair = {}, {}

    def count(label, reason):
        by_label[label] = by_label.get(label, 0) + 1
        by_pair[(label, reason)] = by_pair.get((label, reason), 0) + 1

    for reason, cell in zip(reasons, label_cells):
        labels = _split_object_labels(cell)
        if not labels:
            m = _UNDEFINED_LABEL_REASON.match(str(reason))
            labels = [m.group(1)] if m else ["无标签"]
        for label in labels:
            count(label, reason)
    label_summary = pd.DataFrame([{"标签": k, "数量": v} for k, v in by_label.items()]).sort_values("数量", ascending=False)
    pair_summary = pd.DataFrame([{"标签": k[0], "原因": k[1], "数量": v} for k, v in by_pair.items()]).sort_values("数量", ascending=False)
    return {"reason_summary": reason_counts, "label_summary": label_summary, "reason_label": pair_summary}


def summarize_unclassified(unclassified_excel_path: str, output_dir: str, json_columns: Optional[list] = None):
    """Drop-in for reference processor.py:833-891 (``json_columns`` is accepted and, as there, never used)."""
    if not os.path.exists(unclassified_excel_path):
        raise FileNotFoundError(f"无法分类文件不存在：{unclassified_excel_path}")
    df = pd.read_excel(unclassified_excel_path)
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    sheets = unclassified_summary_frames(df)
    out_path = output_dir / "unclassified_summary.xlsx"
    with pd.ExcelWriter(out_path) as writer:
        for name in ("reason_summary", "label_summary", "reason_label"):
            sheets[name].to_excel(writer, sheet_name=name, index=False)
    return out_path


def _label_file_counts(text: str, names) -> dict:
    """label -> boxes for one label file (reference processor.py:1123-1133): the first field of a line is the class id
    (``int(float(...))``; a line whose id does not parse, or whose lookup fails, is skipped); ids the names list does
    not reach are shown as the number itself.  Negative ids index the list from its end, as there."""
    boxes = {}
    for line in text.splitlines():
        fields = line.strip().split()
        if not fields:
            continue
        try:
            cid = int(float(fields[0]))
            name = names[cid] if cid < len(names) else str(cid)
            boxes[name] = boxes.get(name, 0) + 1
        except Exception:  # noqa: BLE001
            continue
    return boxes


def summarize_yolo_label_counts(dataset_dirs):
    """Drop-in for reference processor.py:1089-1162: for every dataset directory and split, images and boxes per label
    read back from labels/<split>/*.txt (what K7 wrote), with data.yaml's names -> (stats dict, flat DataFrame)."""
    import yaml

    def pct(part, whole):
        return f"{(part / whole * 100):.1f}%" if whole else "0.0%"

    def add(into, counts):
        for k, v in counts.items():
            into[k] = into.get(k, 0) + v

    stats, rows = {}, []
    for entry in dataset_dirs or []:
        if not entry:
            continue
        root = Path(entry)
        if not root.exists():
            continue
        names = []
        yaml_path = root / "data.yaml"
        if yaml_path.exists():
            try:
                names = yaml.safe_load(yaml_path.read_text(encoding="utf-8")).get("names") or []
            except Exception:  # noqa: BLE001
                pass
        splits, all_images, all_img, all_box = {}, 0, {}, {}
        for split in ("train", "val", "test"):
            img_counts, box_counts, images = {}, {}, 0
            folder = root / "labels" / split
            if folder.exists():
                for path in folder.glob("*.txt"):
                    images += 1
                    try:
                        text = path.read_text(encoding="utf-8", errors="ignore")
                    except Exception:  # noqa: BLE001
                        continue
                    in_file = _label_file_counts(text, names)
                    add(box_counts, in_file)
                    add(img_counts, dict.fromkeys(in_file, 1))
            splits[split] = {"total_images": images, "label_counts": img_counts, "box_counts": box_counts}
            all_images += images
            add(all_img, img_counts)
            add(all_box, box_counts)
            rows += [{"数据集": root.name, "split": split, "标签": k, "图片数量": img_counts.get(k, 0), "标注框数量": box_counts.get(k, 0),
                      "占比%": pct(img_counts.get(k, 0), images), "split总图片数": images} for k in set(img_counts) | set(box_counts)]
        splits["all"] = {"total_images": all_images, "label_counts": all_img, "box_counts": all_box}
        stats[root.name] = splits
        rows += [{"数据集": root.name, "split": "all", "标签": k, "图片数量": all_img.get(k, 0), "标注框数量": all_box.get(k, 0),
                  "占比%": pct(all_img.get(k, 0), all_images), "split总图片数": all_images} for k in set(all_img) | set(all_box)]
    return stats, pd.DataFrame(rows)


# =============================================================================== step "download": annotated images
def _annotation_shapes(json_str):
    """(name, [(x, y), ...]) of every object of an annotation cell that can be drawn (reference processor.py:447-462): dict
    objects, the points being the dict entries of polygon.ptList whose x and y are both present and not null; fewer than
    two points: nothing to draw.  Whatever goes wrong while walking the cell ends the walk quietly — the shapes before
    it are kept, as the reference's try / except around its drawing loop keeps what it has drawn."""
    if not isinstance(json_str, str):
        return
    try:
        for obj in json.loads(json_str).get("objects", []):
            if not isinstance(obj, dict):
                continue
            name = obj.get("name", "未知类别")
            points = [(p["x"], p["y"]) for p in obj.get("polygon", {}).get("ptList", [])
                      if isinstance(p, dict) and p.get("x") is not None and p.get("y") is not None]
            if len(points) >= 2:
                yield name, points
    except Exception:  # noqa: BLE001
        return


def _draw_shapes(draw, shapes, colour, font):
    """two points: a rectangle; more: the polygon; the name on a white patch 20 px above the first / top-left point"""
    try:
        for name, points in shapes:
            if len(points) == 2:
                (x1, y1), (x2, y2) = points
                draw.rectangle([x1, y1, x2, y2], outline=colour, width=2)
                anchor = (x1, y1 - 20)
            else:
                draw.polygon(points, outline=colour, width=2)
                anchor = (min(p[0] for p in points), min(p[1] for p in points) - 20)
            draw.rectangle(draw.textbbox(anchor, name, font=font), fill=(255, 255, 255, 180))
            draw.text(anchor, name, font=font, fill=colour)
    except Exception:  # noqa: BLE001
        pass


def download_and_draw_annotations(
        input_csv_path,
        output_dir: Optional[str] = None,
        download_dir: Optional[str] = None,
        result_dir: Optional[str] = None,
        max_images: Optional[int] = None,
        timeout: int = 15
):
    """Drop-in for reference processor.py:409-514 (pipeline step ``download``): every row's image — taken from
    ``download_dir`` when it is there, fetched otherwise — with the original annotation drawn in red and the replaced one
    in green, saved under ``result_dir``.  Host-only (Pillow, requests); rows whose image cannot be had or opened count
    as processed and are skipped; ``max_images`` bounds the rows processed, successful or not."""
    import requests
    from PIL import Image, ImageDraw, ImageFont

    base_dir = Path(output_dir) if output_dir else Path(os.getcwd())
    download_dir = Path(download_dir) if download_dir else base_dir / "downloaded_images"
    result_dir = Path(result_dir) if result_dir else base_dir / "annotated_images"
    download_dir.mkdir(parents=True, exist_ok=True)
    result_dir.mkdir(parents=True, exist_ok=True)
    try:
        df = pd.read_csv(input_csv_path, encoding="utf-8-sig")
    except Exception as e:  # noqa: BLE001
        print(f"读取CSV失败：{e}")
        return
    if not {"source", ANNOTATION_COL, BBOX_COL} <= set(df.columns):
        print("CSV缺少必要列")
        return

    font = None
    for face in ("simhei.ttf", "Arial Unicode.ttf"):                     # :434-441
        try:
            font = ImageFont.truetype(face, 48)
            break
        except Exception:  # noqa: BLE001
            continue
    if font is None:
        font = ImageFont.load_default()

    processed = 0
    for idx, source_url, before, after in zip(df.index, df["source"].tolist(), df[ANNOTATION_COL].tolist(), df[BBOX_COL].tolist()):
        if max_images is not None and processed >= max_images:
            break
        processed += 1                                                   # success or failure, the row counts
        filename = source_url.split("/")[-1] if "/" in source_url else f"image_{idx}.jpg"
        local = download_dir / filename
        if not os.path.exists(local):
            try:
                response = requests.get(source_url, stream=True, timeout=timeout)
                response.raise_for_status()
                with open(local, "wb") as f:
                    for chunk in response.iter_content(chunk_size=8192):
                        f.write(chunk)
            except Exception:  # noqa: BLE001
                continue
        try:
            with Image.open(local) as img:
                draw = ImageDraw.Draw(img)
                _draw_shapes(draw, _annotation_shapes(before), (255, 0, 0), font)
                _draw_shapes(draw, _annotation_shapes(after), (0, 255, 0), font)
                img.save(result_dir / filename)
        except Exception:  # noqa: BLE001
            continue
