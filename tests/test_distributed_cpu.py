"""N>1 path on CPU: world_size-2 (and 3) gloo groups run deal_yolo_daya_amd.distributed with the
oracle as the device stage; the concatenated per-rank results must equal the single-process
answers of pandas / the oracle (masks byte-identical at every world size)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from oracle import lib as olib

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, tmp_path):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), str(port),
                               str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_steps_match_single_process(world, tmp_path):
    res = _run(world, tmp_path)
    rng = np.random.default_rng(123)
    n = 5003
    ids = rng.integers(0, 1800, size=n)
    src = pd.Series([None if k % 97 == 0 else f"http://img.example/{k}.jpg" for k in ids.tolist()], dtype=object)
    ref = pd.Series([f"http://img.example/{k}.jpg" for k in range(0, 1800, 10)] + [None, "nan"], dtype=object)
    cat = rng.integers(-1, 3, size=n).astype(np.int32)

    assert [r["lo"] for r in res] + [res[-1]["hi"]] == [(n * r) // world for r in range(world)] + [n]
    assert all(r["gathered_ok"] for r in res)
    for keep in ("first", "last", False):
        got = np.concatenate([r[f"dedup_{keep}"] for r in res]).astype(bool)
        assert np.array_equal(got, ~src.duplicated(keep=keep).to_numpy()), keep     # pandas = reference's call
    got = np.concatenate([r["ref_hit"] for r in res]).astype(bool)
    want = src.astype(str).isin(set(ref.dropna().astype(str))).to_numpy()
    assert np.array_equal(got, want)

    sizes = np.bincount(cat[cat >= 0], minlength=3).astype(np.int64)
    off = np.zeros(4, np.int64)
    np.cumsum(sizes, out=off[1:])
    perm = np.concatenate([olib.mt19937_permutation(42, int(s)) for s in sizes])
    tr = np.array([int(s * (0.8 / (0.8 + 0.1 + 0.1))) for s in sizes]); va = np.array([int(s * (0.1 / 1.0)) for s in sizes])
    split, pos = olib.split_ids(cat, perm, off, tr, va)
    assert np.array_equal(np.concatenate([r["split"] for r in res]), split)
    assert np.array_equal(np.concatenate([r["pos"] for r in res]), pos)

    boxes = np.round(rng.random((n, 4)) * 500, 1)              # the worker draws the same numbers after `cat`
    boxes[:, 2:] += boxes[:, :2] + 1
    boxes[::50, 2] = boxes[::50, 0]
    off, flag, text = olib.yolo_lines(boxes, np.arange(n + 1, dtype=np.int32), None, np.full(n, 640.0), np.full(n, 480.0),
                                      (np.arange(n) % 13).astype(np.int32))
    assert "".join(r["yolo_text"] for r in res).encode() == text and all(r["yolo_total"] == len(text) for r in res)
    assert np.array_equal(np.concatenate([r["yolo_flag"] for r in res]), flag)
    starts = np.concatenate([np.asarray(r["yolo_off"][:-1]) for r in res] + [[res[-1]["yolo_off"][-1]]])
    assert np.array_equal(starts, off)


@pytest.mark.parametrize("world", [2, 3])
def test_replace_and_iou_shares_are_balanced_by_annotation_bytes(world, tmp_path):
    """SURVEY §8e partitioning: the fused replace -> IoU pass per rank on contiguous shares cut by annotation bytes — the shares'
    frames laid end to end are the single-process frames (row labels included), the totals agree on every rank, and no share
    carries much more than its part of the bytes although the dense images sit in the first 13 % of the rows"""
    from deal_yolo_daya_amd import synth
    from deal_yolo_daya_amd.core import processor as P
    from helpers import OracleBackend

    res = _run(world, tmp_path)
    dense = synth.to_frame(synth.generate(60, seed=31, boxes_per_row=40))
    sparse = synth.to_frame(synth.generate(400, seed=32))
    table = pd.concat([dense, sparse], ignore_index=True)
    table.loc[[5, 200, 459], P.ANNOTATION_COL] = None
    table.index = pd.Index(np.arange(len(table)) * 2 + 1)
    kept, excluded, high, other = P.replace_and_filter_frame(table, 2, 0.98, OracleBackend())
    bounds = [r["rf_bounds"] for r in res]
    assert bounds[0][0] == 0 and bounds[-1][1] == len(table) and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    for k, frame in enumerate((kept, excluded, high, other)):
        assert sum((r["rf_labels"][k] for r in res), []) == frame.index.tolist()
    assert sum((r["rf_bbox"] for r in res), []) == kept[P.BBOX_COL].tolist()
    want = {"rows": len(table), "kept": len(kept), "excluded": len(excluded), "high": len(high), "other": len(other)}
    assert all(r["rf_totals"] == want for r in res) and all(r["rf_per_rank"] == res[0]["rf_per_rank"] for r in res)
    total = res[0]["rf_weight_total"]
    assert sum(r["rf_weight"] for r in res) == total
    heaviest_row = 40 * 12 * 40 + 2000                            # bytes of one dense row, generously
    assert max(r["rf_weight"] for r in res) <= total / world + heaviest_row
    by_rows = [sum(1 for _ in range(*b)) for b in bounds]
    assert by_rows[0] < 0.85 * len(table) / world                 # the first share is short in rows: it holds the dense images


def test_shard_bounds_by_weight_tiles_the_rows():
    from deal_yolo_daya_amd.distributed import shard_bounds, shard_bounds_by_weight
    rng = np.random.default_rng(8)
    for n in (0, 1, 5, 1000):
        for world in (1, 2, 3, 8):
            for w in (rng.integers(0, 50, n), np.zeros(n), np.r_[np.full(n // 2, 1000), np.ones(n - n // 2)]):
                b = [shard_bounds_by_weight(w, world, r) for r in range(world)]
                assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
                assert all(lo <= hi for lo, hi in b)
                if n and w.sum() > 0:
                    share = [w[lo:hi].sum() for lo, hi in b]
                    assert max(share) <= w.sum() / world + w.max()
                elif n:
                    assert b == [shard_bounds(n, world, r) for r in range(world)]


def test_shard_bounds_cover_and_balance():
    from deal_yolo_daya_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
