"""merge_all_csv_in_folder against the reference's own runs (tests/golden/merge_case.json: merged bytes,
progress-callback arguments, printed lines) and, on files larger than one parser read, against the reference's pandas
loop that the product keeps as its fallback.  Host code only: no GPU needed."""
import contextlib
import io
import os
import pathlib

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden
from deal_yolo_daya_amd import synth
from deal_yolo_daya_amd.core import processor as P


@pytest.fixture
def sorted_glob(monkeypatch):
    """directory order is file-system dependent: the fixture was generated with sorted names"""
    orig = pathlib.Path.glob
    monkeypatch.setattr(pathlib.Path, "glob", lambda self, pat: iter(sorted(orig(self, pat))))


def _run(folder, out, chunk_size, tmp):
    calls, buf = [], io.StringIO()
    with contextlib.redirect_stdout(buf):
        ret = P.merge_all_csv_in_folder(folder, out, "utf-8-sig", chunk_size, lambda *a: calls.append(list(a)))
    merged = open(out, "rb").read().decode("utf-8") if os.path.exists(out) else None
    return {"return": ret, "merged": merged, "calls": calls, "printed": buf.getvalue().replace(str(tmp), "<TMP>")}


@pytest.mark.parametrize("run_idx", [0, 1])
def test_merge_matches_reference(tmp_path, sorted_glob, run_idx):
    g = load_golden("merge_case.json")
    folder = tmp_path / "in"
    folder.mkdir()
    for name, text in g["files"].items():
        with open(folder / name, "w", encoding="utf-8-sig", newline="") as f:
            f.write(text)
    want = g["runs"][run_idx]
    got = _run(str(folder), str(tmp_path / "o" / "merged.csv"), want["chunk_size"], tmp_path)
    assert got["merged"] == want["merged"]
    assert got["calls"] == want["calls"]
    assert got["printed"] == want["printed"]
    assert got["return"] == want["return"]
    paths = P.LAST_IO_PATH["merge"]
    assert paths["a_main.csv"] == "native" and paths["b_other_order.csv"] == "native"
    assert paths["c_narrow.csv"] == "pandas" and paths["d_crlf.csv"] == "native" and paths["h_has_source_file.csv"] == "pandas"


def test_merge_progress_offsets_follow_the_parser_reads(tmp_path, monkeypatch, sorted_glob):
    """files of several 262144-character reads with multi-byte text: native path == the reference's pandas loop,
    byte for byte and callback for callback (file_bytes is what f.tell() shows after each chunk)"""
    folder = tmp_path / "in"
    folder.mkdir()
    t = synth.generate(1500, seed=3, max_boxes=6)
    df = synth.to_frame(t)
    df["说明"] = ["中文说明，带逗号" * (1 + i % 5) for i in range(len(df))]
    df["v"] = np.arange(len(df)) * 0.5
    df.to_csv(folder / "big1.csv", index=False, encoding="utf-8-sig")
    df.iloc[::-1].to_csv(folder / "big2.csv", index=False, encoding="utf-8-sig")
    assert os.path.getsize(folder / "big1.csv") > 3 * 262144
    native = _run(str(folder), str(tmp_path / "n.csv"), 150, tmp_path)
    assert set(P.LAST_IO_PATH["merge"].values()) == {"native"}
    monkeypatch.setenv("DYD_NATIVE_CSV", "0")
    plain = _run(str(folder), str(tmp_path / "p.csv"), 150, tmp_path)
    assert set(P.LAST_IO_PATH["merge"].values()) == {"pandas"}
    assert native["merged"] == plain["merged"] and native["return"] == plain["return"] == 3000
    assert native["calls"] == plain["calls"]
    assert len({c[7] for c in native["calls"]}) > 4          # several distinct read positions were reported


def test_merge_errors_and_empty_folder(tmp_path, capsys):
    with pytest.raises(FileNotFoundError):
        P.merge_all_csv_in_folder(str(tmp_path / "nope"))
    (tmp_path / "e").mkdir()
    assert P.merge_all_csv_in_folder(str(tmp_path / "e"), str(tmp_path / "o.csv")) is None
    assert "未找到CSV文件" in capsys.readouterr().out
    (tmp_path / "e" / "x.csv").write_text("")
    assert P.merge_all_csv_in_folder(str(tmp_path / "e"), str(tmp_path / "o.csv")) is None
    assert "没有可合并的有效CSV数据" in capsys.readouterr().out


def test_merge_crlf_files_native_equals_the_pandas_loop(tmp_path, monkeypatch, sorted_glob):
    """CR LF inputs: the text-mode handle delivers "\n" for every CR LF, which moves both the parsed text and the
    f.tell() positions the progress callback reports"""
    folder = tmp_path / "in"
    folder.mkdir()
    df = synth.to_frame(synth.generate(1500, seed=4, max_boxes=6))
    df["说明"] = ["中文说明，带逗号" * (1 + i % 4) for i in range(len(df))]
    df.to_csv(folder / "w1.csv", index=False, encoding="utf-8-sig", lineterminator="\r\n")
    text = df.iloc[::-1].to_csv(index=False)
    lines = text.split("\n")
    with open(folder / "w2_mixed.csv", "w", encoding="utf-8-sig", newline="") as f:
        f.write("".join(ln + ("\r\n" if i % 2 else "\n") for i, ln in enumerate(lines[:-1])))
    native = _run(str(folder), str(tmp_path / "n.csv"), 170, tmp_path)
    assert set(P.LAST_IO_PATH["merge"].values()) == {"native"}
    monkeypatch.setenv("DYD_NATIVE_CSV", "0")
    plain = _run(str(folder), str(tmp_path / "p.csv"), 170, tmp_path)
    assert native["merged"] == plain["merged"] and native["return"] == plain["return"] == 3000
    assert native["calls"] == plain["calls"]
