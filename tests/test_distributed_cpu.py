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


def test_shard_bounds_cover_and_balance():
    from deal_yolo_daya_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
