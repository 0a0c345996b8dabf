"""Parity of the HIP kernels (called through the C ABI) with the CPU oracle and with the golden
outputs of the reference.  Needs a real MI355X: run with ``-m gpu`` on the GPU box.

Bar: bit-exact for every integer / byte / index output (bbox values are selections, so they are
bit-exact too); the only floating-point result, the diagnostic max IoU, within 1e-6.
"""
import numpy as np
import pandas as pd
import pytest

import test_host_steps_cpu as host
from helpers import random_boxes, random_polygons
from oracle import lib as olib

pytestmark = pytest.mark.gpu

IOU_TOL = 1e-6


def same_f64(a, b):
    """bit pattern equality except that any NaN equals any NaN (payload is not a reference output)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint64),
                                                                       b[~np.isnan(b)].view(np.uint64))


def test_device_is_gfx950(native):
    assert "gfx950" in native.device_name()


# ------------------------------------------------------------------------------------- K1
@pytest.fixture(params=[-1, 0, 2], ids=["auto", "tile", "group"])
def k1_variant(request, native):
    """-1 = by the table's shape (the group kernel — sixteen lanes per polygon — from 48 points per polygon on), 0 = the tile
    kernel (a lane per polygon), 2 = the group kernel"""
    native.check(native.lib().dyd_set_option(b"k1_variant", request.param), "set_option")
    yield request.param
    native.check(native.lib().dyd_set_option(b"k1_variant", -1), "set_option")


@pytest.mark.parametrize("n_boxes,max_pts", [(1, 5), (63, 12), (256, 12), (257, 12), (5000, 12), (70000, 40), (300, 700), (1000, 64),
                                             (17, 5000), (4097, 33)])
def test_k1_random(native, k1_variant, n_boxes, max_pts):
    rng = np.random.default_rng(n_boxes * 31 + max_pts)
    xy, off = random_polygons(rng, n_boxes, max_pts)
    box, arg = native.bbox_minmax(xy, off)
    obox, oarg = olib.bbox_minmax(xy, off)
    assert np.array_equal(arg, oarg)
    assert same_f64(box, obox)


def test_k1_group_kernel_ties_across_lanes(native):
    """the group kernel folds sixteen lanes' partial extremes: equal values in different lanes must resolve to the lowest
    index, signed zeros and infinities included, and a NaN must stick only from position 0"""
    native.check(native.lib().dyd_set_option(b"k1_variant", 2), "set_option")
    try:
        rng = np.random.default_rng(5)
        polys = []
        for n in (1, 2, 15, 16, 17, 31, 32, 33, 100, 1000):
            for kind in range(6):
                p = np.round(rng.random((n, 2)) * 4, 0)                       # values 0..4: ties everywhere
                if kind == 1:
                    p[:, 0] = 7.0                                            # one value: index 0 wins both ends
                elif kind == 2:
                    p[rng.integers(0, n, max(1, n // 3)), 0] = -0.0
                    p[rng.integers(0, n, max(1, n // 3)), 1] = 0.0
                elif kind == 3:
                    p[0] = np.nan                                            # poisons both coordinates
                elif kind == 4:
                    p[rng.integers(0, n, max(1, n // 2))] = np.nan           # ignored unless at position 0
                elif kind == 5:
                    p[rng.integers(0, n, max(1, n // 4)), 0] = np.inf
                    p[rng.integers(0, n, max(1, n // 4)), 1] = -np.inf
                polys.append(p)
        off = np.zeros(len(polys) + 1, np.int32)
        np.cumsum([len(p) for p in polys], out=off[1:])
        xy = np.concatenate(polys)
        box, arg = native.bbox_minmax(xy, off)
        obox, oarg = olib.bbox_minmax(xy, off)
        assert np.array_equal(arg, oarg) and same_f64(box, obox)
        assert np.array_equal(np.signbit(box), np.signbit(obox))
    finally:
        native.check(native.lib().dyd_set_option(b"k1_variant", -1), "set_option")


def test_k1_edges(native, k1_variant):
    # empty input, all-empty boxes, one giant box spanning many LDS chunks, NaN / -0.0 / tie rules
    box, arg = native.bbox_minmax(np.zeros((0, 2)), np.zeros(1, np.int32))
    assert box.shape == (0, 4) and arg.shape == (0, 4)
    box, arg = native.bbox_minmax(np.zeros((0, 2)), np.zeros(301, np.int32))
    assert np.isnan(box).all() and (arg == -1).all()
    rng = np.random.default_rng(3)
    xy = np.round(rng.random((20000, 2)) * 1000, 0)
    off = np.array([0, 3, 3, 19000, 20000], np.int32)
    box, arg = native.bbox_minmax(xy, off)
    obox, oarg = olib.bbox_minmax(xy, off)
    assert np.array_equal(arg, oarg) and same_f64(box, obox)
    xy = np.array([[0.0, -0.0], [-0.0, 0.0], [0.0, 0.0],            # signed-zero ties: first wins
                   [np.nan, 1], [2, np.nan], [3, 0],                # NaN first poisons x; later NaN y ignored
                   [2, 5], [np.nan, np.nan], [3, 4]], np.float64)   # NaN in the middle ignored
    off = np.array([0, 3, 6, 9], np.int32)
    box, arg = native.bbox_minmax(xy, off)
    assert arg.tolist() == [[0, 0, 0, 0], [0, 2, 0, 0], [0, 2, 2, 0]]
    assert np.signbit(box[0]).tolist() == [False, True, False, True]
    assert np.isnan(box[1, 0]) and np.isnan(box[1, 2]) and box[1, 1] == 0 and box[1, 3] == 1
    assert box[2].tolist() == [2, 4, 3, 5]


def test_k1_direct_variant_agrees(native):
    rng = np.random.default_rng(11)
    xy, off = random_polygons(rng, 4000, 20)
    ref = native.bbox_minmax(xy, off)
    native.check(native.lib().dyd_set_option(b"k1_variant", 1), "set_option")
    try:
        alt = native.bbox_minmax(xy, off)
    finally:
        native.check(native.lib().dyd_set_option(b"k1_variant", -1), "set_option")
    assert np.array_equal(ref[1], alt[1]) and same_f64(ref[0], alt[0])


# ------------------------------------------------------------------------------------- K2
@pytest.mark.parametrize("n_rows,max_boxes,fixed", [(1, 4, None), (31, 32, None), (32, 32, None), (33, 32, None),
                                                    (2000, 32, None), (500, 80, None), (40, None, 256),
                                                    (3, None, 1500), (5, None, 2500)])
@pytest.mark.parametrize("thr,min_boxes", [(0.98, 2), (0.5, 3), (0.0, 2), (1.0, 2)])
def test_k2_random(native, n_rows, max_boxes, fixed, thr, min_boxes):
    rng = np.random.default_rng(n_rows * 7 + (fixed or 0))
    box, off = random_boxes(rng, n_rows, max_boxes or 1, fixed=fixed)
    got = native.iou_any_ge(box, off, min_boxes, thr)
    want = olib.iou_any_ge(box, off, min_boxes, thr)
    assert np.array_equal(got, want)


def test_k2_max_iou_diagnostic(native):
    rng = np.random.default_rng(5)
    box, off = random_boxes(rng, 600, 40, special=False)
    got, gmx = native.iou_any_ge(box, off, 2, 0.9, want_max=True)
    want, wmx = olib.iou_any_ge(box, off, 2, 0.9, want_max=True)
    assert np.array_equal(got, want)
    assert np.max(np.abs(gmx - wmx)) <= IOU_TOL          # the float tolerance north_star states
    assert np.array_equal(gmx.view(np.uint64), wmx.view(np.uint64))   # and in fact the same bits


def test_k2_edges(native):
    assert native.iou_any_ge(np.zeros((0, 4)), np.zeros(1, np.int32), 2, 0.98).shape == (0,)
    assert native.iou_any_ge(np.zeros((0, 4)), np.zeros(100, np.int32), 2, 0.98).tolist() == [0] * 99
    tie = np.array([[0, 0, 100, 100], [0, 0, 100, 98]], np.float64)   # IoU == 0.98 exactly -> HIGH
    assert native.iou_any_ge(tie, np.array([0, 2], np.int32), 2, 0.98).tolist() == [1]
    assert native.iou_any_ge(tie, np.array([0, 2], np.int32), 2, 0.9800000000000001).tolist() == [0]
    assert native.iou_any_ge(tie, np.array([0, 2], np.int32), 3, 0.98).tolist() == [0]
    nan_first = np.array([[np.nan, 0, 10, 10], [0, 0, 10, 10]], np.float64)
    nan_second = nan_first[::-1].copy()
    for b in (nan_first, nan_second):
        for thr in (0.98, 0.0, -1.0):
            assert np.array_equal(native.iou_any_ge(b, np.array([0, 2], np.int32), 2, thr),
                                  olib.iou_any_ge(b, np.array([0, 2], np.int32), 2, thr))


# ------------------------------------------------------------------------------------- K3 / K4 / K5
def _strings(rng, n, dup_frac=0.4):
    ids = rng.integers(0, max(1, int(n * (1 - dup_frac))), size=n)
    vals = [f"http://img.example/{k}.jpg" if k % 7 else "x" * int(k % 61) for k in ids.tolist()]
    enc = [v.encode() for v in vals]
    off = np.zeros(n + 1, np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    return np.frombuffer(b"".join(enc), np.uint8), off, vals


@pytest.mark.parametrize("n", [1, 255, 256, 257, 10000, 300000])
def test_k3_hash(native, n):
    data, off, _ = _strings(np.random.default_rng(n), n)
    assert np.array_equal(native.hash128(data, off), olib.hash128(data, off))


def test_k3_known_answers_and_lengths(native):
    def h(b):
        return tuple(int(v) for v in native.hash128(np.frombuffer(b, np.uint8), np.array([0, len(b)], np.int64))[0])
    assert h(b"") == (0, 0)
    assert h(b"hello") == (0xcbd8a7b341bd9b02, 0x5b1e906a48ae1d19)
    blob = bytes(range(256)) * 2
    lens = list(range(0, 70)) + [127, 128, 129, 255]
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    data = np.frombuffer((blob * 8)[: int(off[-1])], np.uint8)
    assert np.array_equal(native.hash128(data, off), olib.hash128(data, off))


@pytest.mark.parametrize("keep", ["first", "last", False])
@pytest.mark.parametrize("n", [1, 2, 1000, 200000])
def test_k4_dedup(native, n, keep):
    data, off, vals = _strings(np.random.default_rng(n + 1), n)
    h = native.hash128(data, off)
    got = native.dedup(h, keep)
    assert np.array_equal(got, olib.dedup(h, {"first": 0, "last": 1, False: 2}[keep]))
    want = ~pd.Series(vals).duplicated(keep=keep).to_numpy()          # pandas = the reference's own call
    assert np.array_equal(got.astype(bool), want)


def test_k4_adversarial_slots(native):
    """many keys sharing the low hash bits (same home slot) and h1 collisions with different h2"""
    n = 5000
    h = np.zeros((n, 2), np.uint64)
    h[:, 0] = np.uint64(12345) + (np.arange(n, dtype=np.uint64) % np.uint64(3)) * np.uint64(1 << 40)
    h[:, 1] = np.arange(n, dtype=np.uint64) % np.uint64(1700)
    for keep, mode in (("first", 0), ("last", 1), (False, 2)):
        assert np.array_equal(native.dedup(h, keep), olib.dedup(h, mode))


@pytest.mark.parametrize("n,r", [(1000, 0), (1000, 1), (50000, 7000), (10, 100000)])
def test_k5_isin(native, n, r):
    rng = np.random.default_rng(n + r)
    h = rng.integers(0, 2 ** 63, size=(n, 2), dtype=np.uint64)
    ref = rng.integers(0, 2 ** 63, size=(r, 2), dtype=np.uint64)
    if r and n:
        take = rng.integers(0, r, size=n // 3)
        h[: len(take)] = ref[take]
    assert np.array_equal(native.isin(h, ref), olib.isin(h, ref))


# ------------------------------------------------------------------------------------- K6
@pytest.mark.parametrize("n,n_cat", [(1, 1), (5000, 2), (100000, 3), (70000, 1500), (4097, 40)])
def test_k6_split(native, n, n_cat):
    rng = np.random.default_rng(n + n_cat)
    cat = rng.integers(-1, n_cat, size=n).astype(np.int32)
    sizes = np.bincount(cat[cat >= 0], minlength=n_cat).astype(np.int64)
    cat_off = np.zeros(n_cat + 1, np.int64)
    np.cumsum(sizes, out=cat_off[1:])
    perm = np.concatenate([native.mt19937_permutation(42, int(s)) for s in sizes])
    n_train = (sizes * 0.8).astype(np.int64)
    n_val = (sizes * 0.1).astype(np.int64)
    got = native.split_ids(cat, perm, cat_off, n_train, n_val)
    want = olib.split_ids(cat, perm, cat_off, n_train, n_val)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    for c in range(min(n_cat, 5)):                       # positions inside a category are a permutation
        assert np.array_equal(np.sort(got[1][cat == c]), np.arange(sizes[c]))


def test_permutation_matches_numpy(native):
    for seed, n in ((42, 100000), (0, 7), (123456, 65537)):
        assert np.array_equal(native.mt19937_permutation(seed, n), np.random.RandomState(seed).permutation(n))


# ------------------------------------------------------------------------------------- golden, through the steps
def test_replace_golden_gpu(native, tmp_path):
    host.run_replace_golden(native, tmp_path)


def test_iou_golden_gpu(native, tmp_path):
    host.run_iou_golden(native, tmp_path)


def test_dedup_golden_gpu(native, tmp_path):
    host.run_dedup_golden(native, tmp_path)


def test_ref_filter_golden_gpu(native, tmp_path):
    host.run_ref_filter_golden(native, tmp_path)


def test_split_golden_gpu(native):
    host.run_split_golden(native)


def test_chain_golden_gpu(native, tmp_path):
    """replace -> IoU through the fused K1+K2 launch behind the step API (process_csv_replace_and_filter,
    replace_and_filter_frame): the reference's files of both steps, byte for byte"""
    host.run_chain_golden(native, tmp_path)


def test_e2e_golden_gpu(native, tmp_path):
    host.run_e2e_golden(native, tmp_path)


def test_default_backend_is_native(native):
    from deal_yolo_daya_amd.backend import default_backend
    assert default_backend() is native
    from deal_yolo_daya_amd.core import processor as P
    assert P.dedup_keep_mask(pd.Series(["a", "b", "a", None, None])).tolist() == [True, True, False, True, False]


# ------------------------------------------------------------------------------------- multi-GPU pieces on one GPU
def test_sharded_device_stage_one_process(native):
    """The device side of the sharded path (global-key table, rank-based split) with the
    all-gather emulated in-process: G shards of one table must reproduce the unsharded masks."""
    import torch
    from deal_yolo_daya_amd import distributed as D
    from deal_yolo_daya_amd import flatten

    ops = D.HipOps("cuda:0")
    rng = np.random.default_rng(77)
    n, G = 40007, 4
    ids = rng.integers(0, 15000, size=n)
    src = pd.Series([None if k % 211 == 0 else f"http://img.example/{k}.jpg" for k in ids.tolist()], dtype=object)
    bounds = [D.shard_bounds(n, G, r) for r in range(G)]
    keys = [D._local_keys(src.iloc[lo:hi], ops) for lo, hi in bounds]
    all_h = torch.cat(keys)
    for keep in ("first", "last", False):
        got = np.concatenate([ops.dedup_global(all_h, lo, hi - lo, keep).cpu().numpy() for lo, hi in bounds])
        assert np.array_equal(got.astype(bool), ~src.duplicated(keep=keep).to_numpy()), keep
    cat = rng.integers(-1, 5, size=n).astype(np.int32)
    sizes = np.bincount(cat[cat >= 0], minlength=5).astype(np.int64)
    off = np.zeros(6, np.int64)
    np.cumsum(sizes, out=off[1:])
    perm = np.concatenate([native.mt19937_permutation(7, int(s)) for s in sizes])
    tr, va = (sizes * 0.8).astype(np.int64), (sizes * 0.1).astype(np.int64)
    want = olib.split_ids(cat, perm, off, tr, va)
    got_s, got_p = [], []
    for lo, hi in bounds:
        base = np.bincount(cat[:lo][cat[:lo] >= 0], minlength=5).astype(np.int64)
        s, p = ops.split_ids_sharded(ops.tensor(cat[lo:hi]), ops.tensor(perm), ops.tensor(off), ops.tensor(tr),
                                     ops.tensor(va), ops.tensor(base))
        got_s.append(s.cpu().numpy()); got_p.append(p.cpu().numpy())
    assert np.array_equal(np.concatenate(got_s), want[0]) and np.array_equal(np.concatenate(got_p), want[1])


def test_sharded_functions_world1_rccl(native):
    """The collective plumbing on the GPU: a world_size-1 RCCL group (one rank per GPU)."""
    import torch
    import torch.distributed as dist
    from deal_yolo_daya_amd import distributed as D

    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1,
                                device_id=torch.device("cuda:0"))
    try:
        ops = D.HipOps("cuda:0")
        src = pd.Series(["a", "b", "a", np.nan, "c", np.nan, "b"], dtype=object)
        ref = pd.Series(["b", "nan", np.nan], dtype=object)
        assert D.dedup_keep_mask_sharded(src, "first", ops).tolist() == [True, True, False, True, True, False, False]
        want = src.astype(str).isin(set(ref.dropna().astype(str))).tolist()     # NaN -> "nan" hits the literal
        assert want == [False, True, False, True, False, True, True]
        assert D.ref_hit_mask_sharded(src, ref, ops).tolist() == want
        split, pos = D.split_ids_sharded(np.array([0, 1, 0, -1, 0, 1], np.int32), 2, ops=ops)
        assert split[3] == 255 and sorted(pos[[0, 2, 4]].tolist()) == [0, 1, 2]
        rng = np.random.default_rng(3)
        box = np.round(rng.random((3000, 4)) * 400, 1)
        box[:, 2:] += box[:, :2] + 1
        box[::40, 3] = box[::40, 1]
        row_off = np.arange(3001, dtype=np.int32)
        w, h, cid = np.full(3000, 640.0), np.full(3000, 480.0), (np.arange(3000) % 17).astype(np.int32)
        goff, gflag, gtext, gtotal = D.yolo_lines_sharded(box, row_off, None, w, h, cid, ops)
        ooff, oflag, otext = olib.yolo_lines(box, row_off, None, w, h, cid)
        assert gtext == otext and gtotal == len(otext) and np.array_equal(goff, ooff) and np.array_equal(gflag, oflag)
    finally:
        dist.destroy_process_group()


def test_whole_chain_with_label_replace_matches_the_cpu_port(native, tmp_path, monkeypatch):
    """dedup -> ref filter -> replace -> IoU -> label_replace -> split -> label texts, CSV in / CSV out where the
    reference has files, product (HIP + native host code) against the CPU port step by step on one synthetic table"""
    from deal_yolo_daya_amd import synth
    from deal_yolo_daya_amd.core import processor as P
    from oracle import steps as osteps

    n = 1500
    df = synth.to_frame(synth.generate(n, seed=77, max_boxes=8))
    df = pd.concat([df, df.iloc[::7]], ignore_index=True)                      # duplicates for the dedup step
    mapping = pd.DataFrame({"old": [f"c{i}" for i in range(0, 20, 3)], "new": ["c1", "c2", "c1", "c30", "c4", "c2", "c1"]})
    monkeypatch.setattr(pd, "read_excel", lambda *a, **k: mapping.copy())
    monkeypatch.setattr(pd.DataFrame, "to_excel", lambda self, *a, **k: None)
    Q = lambda name: str(tmp_path / name)  # noqa: E731
    df.to_csv(Q("in.csv"), index=False, encoding="utf-8-sig")
    pd.DataFrame({"source": synth.reference_urls(n)}).to_csv(Q("ref.csv"), index=False, encoding="utf-8-sig")

    def same(a, b):
        return open(Q(a), "rb").read() == open(Q(b), "rb").read()

    P.deduplicate_csv_by_source(Q("in.csv"), Q("p1.csv"), verbose=False)
    osteps.dedup_csv(Q("in.csv"), Q("o1.csv"))
    assert same("p1.csv", "o1.csv")
    P.remove_duplicates_between_csv(Q("p1.csv"), Q("ref.csv"), Q("p2.csv"), verbose=False)
    osteps.ref_filter_csv(Q("o1.csv"), Q("ref.csv"), Q("o2.csv"))
    assert same("p2.csv", "o2.csv")
    P.process_csv_replace_ptlist(Q("p2.csv"), Q("p3.csv"), Q("p3e.csv"))
    osteps.replace_csv(Q("o2.csv"), Q("o3.csv"), Q("o3e.csv"))
    assert same("p3.csv", "o3.csv") and same("p3e.csv", "o3e.csv")
    P.filter_by_box_count_and_iou(Q("p3.csv"), Q("p4h.csv"), Q("p4o.csv"), 2, 0.98)
    osteps.iou_filter_csv(Q("o3.csv"), Q("o4h.csv"), Q("o4o.csv"), 2, 0.98)
    assert same("p4h.csv", "o4h.csv") and same("p4o.csv", "o4o.csv")
    res = P.replace_labels_by_mapping(Q("p4o.csv"), "map.xlsx", Q("p5.csv"), diff_excel_path=Q("d.xlsx"), unmatched_excel_path=Q("u.xlsx"))
    want = osteps.label_replace_csv(Q("o4o.csv"), mapping, Q("o5.csv"), diff_excel_path="d", unmatched_excel_path="u")
    assert same("p5.csv", "o5.csv") and res["summary"] == want["summary"] and res["sample_diff"] == want["sample_diff"]
    assert res["summary"]["replaced_labels"] > 1000 and P.LAST_IO_PATH["label_replace"] == "native"

    table = pd.read_csv(Q("p5.csv"), encoding="utf-8-sig")
    rules = pd.DataFrame({"catA": [f"c{i}" for i in range(1, 9)], "catB": [f"c{i}" for i in range(10, 18)]})
    lmap = P.rules_to_label_map(rules)
    got, exp = P.split_frames(table, lmap), osteps.split_frames(table, osteps.rules_to_map(rules))
    assert list(got["categories"]) == list(exp["categories"])
    for cat in exp["categories"]:
        for a, b in zip(got["categories"][cat], exp["categories"][cat]):
            assert a.equals(b)
    assert got["unclassified"].equals(exp["unclassified"]) and got["split_counts"].equals(exp["split_counts"])
    sheet = got["categories"]["catA"][0]
    labels = sheet["分类标签"].tolist()
    classes = sorted(set(labels))
    cids = [classes.index(v) for v in labels]
    texts, _ = P.yolo_label_texts(sheet[P.BBOX_COL].tolist(), labels, cids, sheet["width"].tolist(), sheet["height"].tolist())
    assert texts == [osteps.yolo_row_text(c, lab, k, w, h)[0] for c, lab, k, w, h in
                     zip(sheet[P.BBOX_COL], labels, cids, sheet["width"], sheet["height"])] and any(texts)


@pytest.mark.parametrize("sizes", [[1025], [3000, 5, 0, 1200], [10, 4100, 2, 1030, 7], [257, 256, 300, 1, 600], [300] * 2100 + [900],
                                   [512, 513, 1023, 1024, 1025, 511, 64, 5, 768]])
@pytest.mark.parametrize("thr,min_boxes", [(0.98, 2), (0.3, 2), (0.0, 2), (0.98, 5000)])
def test_k2_rows_of_thousands_of_boxes(native, sizes, thr, min_boxes):
    """rows above 256 boxes leave the main kernel for k2_big_rows_kernel (up to 1024 boxes: sorted and swept by one wave, 8 or 16 keys per
    lane; beyond: all pairs spread over the grid): same flags and the same maximum IoU as the oracle's double loop, NaN corners and exact
    ties included"""
    rng = np.random.default_rng(sum(sizes) + int(thr * 100))
    off = np.zeros(len(sizes) + 1, np.int32)
    np.cumsum(sizes, out=off[1:])
    nb = int(off[-1])
    c = np.round(rng.random((nb, 2)) * 3000, 0)
    box = np.concatenate([c, c + np.round(rng.random((nb, 2)) * 80 + 1, 0)], axis=1)
    box[rng.integers(0, nb, 40), rng.integers(0, 4, 40)] = np.nan
    swap = rng.random(nb) < 0.3
    box[swap] = box[swap][:, [2, 3, 0, 1]]
    for r in range(len(sizes)):                                   # a near-duplicate pair deep inside every big row
        if sizes[r] > 256:
            a, b = off[r] + sizes[r] - 3, off[r] + sizes[r] // 2
            box[a] = [10, 10, 110, 110]
            box[b] = [10, 10, 110, 109] if r % 2 == 0 else [500, 500, 501, 501]
    got, gmx = native.iou_any_ge(box, off, min_boxes, thr, want_max=True)
    want, wmx = olib.iou_any_ge(box, off, min_boxes, thr, want_max=True)
    assert np.array_equal(got, want)
    assert np.array_equal(gmx.view(np.uint64), wmx.view(np.uint64))
    assert np.array_equal(native.iou_any_ge(box, off, min_boxes, thr), want)


def test_a_full_hash_table_is_an_error_not_a_wrong_mask(native):
    """K4 / K5 record device-side failures in the status word: the host-pointer entries return an error, a `_dev` caller
    finds it in dyd_device_status (csrc/k4_dedup.hip: table full).  dyd_set_option("k4_capacity_shift") undersizes the table."""
    import torch

    L = native.lib()
    rng = np.random.default_rng(3)
    h = rng.integers(0, 2 ** 63, size=(50_000, 2), dtype=np.uint64)
    native.check(L.dyd_set_option(b"k4_capacity_shift", 3), "opt")          # 131072 slots / 8 = 16384 < 50000 keys
    try:
        with pytest.raises(native.NativeError, match="hash table full"):
            native.dedup(h, "first")
        with pytest.raises(native.NativeError, match="hash table full"):
            native.isin(h[:100], h)
        t = torch.from_numpy(h.view(np.int64)).to("cuda:0")
        keep = torch.empty(len(h), dtype=torch.uint8, device="cuda:0")
        native.check(L.dyd_dedup_dev(t.data_ptr(), len(h), 0, keep.data_ptr(), None), "dyd_dedup_dev")    # queued: no error yet
        assert L.dyd_device_status(None) != 0 and b"hash table full" in L.dyd_last_error()
        assert L.dyd_device_status(None) == 0                                                          # read once, then cleared
    finally:
        native.check(L.dyd_set_option(b"k4_capacity_shift", 0), "opt")
    assert np.array_equal(native.dedup(h, "first"), olib.dedup(h, 0))
    assert L.dyd_device_status(None) == 0


def test_native_pipeline_equals_the_stepwise_route(native, monkeypatch):
    """replace_and_filter_frame through dyd_json_replace_iou (per-part scan / fused launch / emit) and with DYD_NATIVE_PIPELINE=0
    (scan -> one gathered launch -> emit) on fuzzed cells incl. irregular, undecodable, NaN and big-int ones: identical frames"""
    import json as _json
    import random

    import pandas as pd
    from test_native_json_cpu import Gen
    from deal_yolo_daya_amd import synth
    from deal_yolo_daya_amd.core import processor as P
    from oracle import steps as osteps

    cells = []
    for s in range(4000):
        try:
            c = Gen(5000 + s).cell()
            out = osteps.replace_cell(c)              # keep the cells both steps survive (the raising ones are tested elsewhere)
            if out is not None:
                osteps.row_is_high(osteps.boxes_of_cell(out), 2, 2.0)      # thr 2.0: no early exit, every pair is evaluated
            cells.append(c)
        except Exception:  # noqa: BLE001
            pass
    t = synth.generate(3000, seed=9)
    cells += synth.json_cells(t).tolist()
    cells += [None, float("nan"), 7, "", " "]
    random.Random(1).shuffle(cells)
    df = pd.DataFrame({"source": [f"u{i}" for i in range(len(cells))], P.ANNOTATION_COL: pd.Series(cells, dtype=object)})
    for mb, thr in ((2, 0.98), (3, 0.5)):
        stats = {}
        a = P.replace_and_filter_frame(df, mb, thr, stats=stats)
        assert stats.get("native_pipeline", 0) >= 1
        monkeypatch.setenv("DYD_NATIVE_PIPELINE", "0")
        stats2 = {}
        b = P.replace_and_filter_frame(df, mb, thr, stats=stats2)
        monkeypatch.delenv("DYD_NATIVE_PIPELINE")
        assert "native_pipeline" not in stats2
        for x, y in zip(a, b):
            pd.testing.assert_frame_equal(x, y)
        okept, oproj, _ = osteps.replace_frame(df)
        ohi, _ = osteps.iou_filter_frame(oproj, mb, thr)
        assert a[0][P.BBOX_COL].tolist() == okept[osteps.NEW_COL].tolist()
        assert a[2]["source"].tolist() == ohi["source"].tolist()
    # the pass works chunk by chunk through a staging slot: chunks of one cell, of a few cells, and of everything give the same frames
    base = P.replace_and_filter_frame(df, 2, 0.98)
    for kb in ("1", "64", "1000000"):
        monkeypatch.setenv("DYD_PIPE_CHUNK_KB", kb)
        stats = {}
        c = P.replace_and_filter_frame(df, 2, 0.98, stats=stats)
        assert stats.get("native_pipeline", 0) >= 1
        for x, y in zip(c, base):
            pd.testing.assert_frame_equal(x, y)
    monkeypatch.delenv("DYD_PIPE_CHUNK_KB")


def test_staged_fused_entry_with_and_without_pinned_memory(native):
    """dyd_bbox_iou_fused_staged through one slot, call after call (arenas grow and are kept), from pageable arrays and from the
    slot's own pinned arena: the flags and arg indices of the plain host entry"""
    import ctypes as C
    from deal_yolo_daya_amd import synth
    L = native.lib()
    t = synth.generate(4000, seed=21)
    want_arg, want_high = native.bbox_iou_fused(t.xy, t.pt_off, t.box_off, 2, 0.98)
    stage, pin, cap = C.c_void_p(), C.c_void_p(), C.c_size_t()
    native.check(L.dyd_stage_acquire(t.xy.nbytes + 4096, C.byref(stage), C.byref(pin), C.byref(cap)), "dyd_stage_acquire")
    try:
        assert cap.value == 0 or cap.value >= t.xy.nbytes
        for rows in (4000, 700, 4000, 1):
            nb = int(t.box_off[rows]); npts = int(t.pt_off[nb])
            arg = np.full((nb, 4), -9, np.int32); high = np.full(rows, 9, np.uint8)
            xy = t.xy[:npts]
            if pin.value and rows != 700:                      # the points inside the slot's pinned arena
                xy = np.ctypeslib.as_array(C.cast(pin.value, C.POINTER(C.c_double)), shape=(npts, 2))
                xy[:] = t.xy[:npts]
            po = np.ascontiguousarray(t.pt_off[:nb + 1]); bo = np.ascontiguousarray(t.box_off[:rows + 1])
            native.check(L.dyd_bbox_iou_fused_staged(stage, xy.ctypes.data, po.ctypes.data, bo.ctypes.data, rows, 2, 0.98, None,
                                                     arg.ctypes.data, high.ctypes.data), "dyd_bbox_iou_fused_staged")
            assert np.array_equal(arg, want_arg[:nb]) and np.array_equal(high, want_high[:rows])
    finally:
        L.dyd_stage_release(stage)


def test_arrow_backed_bbox_column_holds_the_same_values(native):
    """replace_and_filter_frame(text_dtype="arrow"): the bbox column laid over the emitter's buffers (pandas string dtype,
    pyarrow storage) equals the object column value for value, survives the function's locals, and writes the same CSV"""
    import gc

    import pandas as pd
    from deal_yolo_daya_amd import synth
    from deal_yolo_daya_amd.core import processor as P

    t = synth.generate(5000, seed=21)
    cells = synth.json_cells(t)
    cells[17], cells[4000] = '{"objects": [', None              # an undecodable cell and a missing one
    df = pd.DataFrame({"source": synth.urls(t), P.ANNOTATION_COL: cells})
    ko, _, ho, _ = P.replace_and_filter_frame(df, 2, 0.98)
    ka, _, ha, oa = P.replace_and_filter_frame(df, 2, 0.98, text_dtype="arrow")
    gc.collect()
    assert str(ka[P.BBOX_COL].dtype) == "string" and ko[P.BBOX_COL].dtype == object
    assert ka[P.BBOX_COL].isna().tolist() == ko[P.BBOX_COL].isna().tolist() and int(ka[P.BBOX_COL].isna().sum()) == 1
    assert ka[P.BBOX_COL].dropna().tolist() == ko[P.BBOX_COL].dropna().tolist()
    assert ha["source"].tolist() == ho["source"].tolist() and len(ha) + len(oa) == len(ka)
    assert ka.to_csv(index=False) == ko.to_csv(index=False)


def test_partner_queries_and_the_default_verification(native):
    """dyd_dedup_partner / dyd_isin_partner (who was matched to whom) against numpy, and the step functions' default verification on
    the device path: a doctored hash that makes two different URLs collide must not cost a row"""
    import pandas as pd
    from helpers import OracleBackend
    from deal_yolo_daya_amd.core import processor as P
    rng = np.random.default_rng(77)
    for n in (1, 5, 4097, 300_000):
        h = rng.integers(0, 2 ** 63, size=(n, 2)).astype(np.uint64)
        h[rng.integers(0, n, n // 2)] = h[rng.integers(0, n, n // 2)]                  # many duplicates
        h[:, 0] &= np.uint64(0xffff) if n > 1000 else np.uint64(0xffffffffffffffff)       # crowded slots
        assert np.array_equal(native.dedup_partner(h), OracleBackend().dedup_partner(h)), n
        ref = np.concatenate([h[rng.integers(0, n, max(1, n // 3))], rng.integers(0, 2 ** 63, size=(7, 2)).astype(np.uint64)])
        got = native.isin_partner(h, ref)
        hit = native.isin(h, ref).astype(bool)
        assert np.array_equal(got >= 0, hit) and np.array_equal(ref[got[hit]], h[hit]), n
    assert native.isin_partner(h, np.zeros((0, 2), np.uint64)).tolist() == [-1] * len(h)

    class Colliding:
        def __getattr__(self, name):
            return getattr(native, name)

        def hash128(self, data, off):
            out = native.hash128(data, off)
            if len(out) > 2:                      # the main column only: rows 0 and 1 collide
                out[1] = out[0]
            return out

    urls = pd.Series([f"http://img.example/{k % 5000}.jpg" for k in range(20_000)] + [None], name="source")
    P.VERIFY_EVENTS.clear()
    assert np.array_equal(P.dedup_keep_mask(urls, "first"), ~urls.duplicated(keep="first").to_numpy()) and not P.VERIFY_EVENTS
    assert not P.dedup_keep_mask(urls, "first", Colliding(), verify=False)[1]
    assert np.array_equal(P.dedup_keep_mask(urls, "first", Colliding()), ~urls.duplicated(keep="first").to_numpy()) and len(P.VERIFY_EVENTS) == 1
    ref = pd.Series(["http://img.example/0.jpg", "http://img.example/77.jpg"], name="source")
    want = urls.astype(str).isin(set(ref.astype(str))).to_numpy()
    assert np.array_equal(P.ref_hit_mask(urls, ref), want) and len(P.VERIFY_EVENTS) == 1
    assert np.array_equal(P.ref_hit_mask(urls, ref, Colliding()), want) and len(P.VERIFY_EVENTS) == 2
    P.VERIFY_EVENTS.clear()
