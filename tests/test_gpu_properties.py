"""Size-independent properties of the device stage at BASELINE's bench size (1M rows, ~16.5M
boxes, ~124M points: configs[1]), where a full oracle run is still cheap enough to use as well.
Needs a real MI355X (-m gpu)."""
import numpy as np
import pytest

from oracle import lib as olib

pytestmark = pytest.mark.gpu

ROWS = 1_000_000


@pytest.fixture(scope="module")
def table():
    from deal_yolo_daya_amd import synth
    return synth.generate(ROWS, seed=synth.SEED)


@pytest.fixture(scope="module")
def device_result(native, table):
    """One fused launch over the whole table through the _dev ABI; results copied back."""
    import torch
    dev = torch.device("cuda:0")
    L = native.lib()
    xy = torch.from_numpy(table.xy).to(dev)
    po, bo = torch.from_numpy(table.pt_off).to(dev), torch.from_numpy(table.box_off).to(dev)
    B, N = table.n_boxes, table.n_rows
    box = torch.empty((B, 4), dtype=torch.float64, device=dev)
    arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
    high = torch.empty(N, dtype=torch.uint8, device=dev)
    native.check(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), po.data_ptr(), bo.data_ptr(), N, B, int(xy.shape[0]), 2, 0.98, box.data_ptr(),
                                          arg.data_ptr(), high.data_ptr(), torch.cuda.current_stream().cuda_stream), "fused")
    torch.cuda.synchronize()
    return box.cpu().numpy(), arg.cpu().numpy(), high.cpu().numpy()


def test_full_size_matches_oracle(table, device_result):
    box, arg, high = device_result
    obox, oarg = olib.bbox_minmax(table.xy, table.pt_off)
    assert np.array_equal(arg, oarg)
    assert np.array_equal(box.view(np.uint64), obox.view(np.uint64))
    assert np.array_equal(high, olib.iou_any_ge(obox, table.box_off, 2, 0.98))


def test_bbox_properties(table, device_result):
    box, arg, _ = device_result
    xy, off = table.xy, table.pt_off.astype(np.int64)
    starts = off[:-1]
    # the winning index really holds the reported value (selection, not arithmetic)
    assert np.array_equal(xy[starts + arg[:, 0], 0], box[:, 0]) and np.array_equal(xy[starts + arg[:, 1], 1], box[:, 1])
    assert np.array_equal(xy[starts + arg[:, 2], 0], box[:, 2]) and np.array_equal(xy[starts + arg[:, 3], 1], box[:, 3])
    # extremal: equals numpy's segmented min / max
    assert np.array_equal(np.minimum.reduceat(xy[:, 0], starts), box[:, 0])
    assert np.array_equal(np.maximum.reduceat(xy[:, 1], starts), box[:, 3])
    # first-wins: no earlier point of the box carries the same extreme
    k = np.flatnonzero(arg[:, 0] > 0)[:200000]
    for j in (1, 2, 3):
        prev = xy[starts[k] + arg[k, 0] - j, 0]
        ok = (arg[k, 0] - j < 0) | (prev != box[k, 0])
        assert ok.all()


def test_bbox_is_idempotent(native, device_result):
    """the bbox of a bbox's two corner points is the bbox itself (the IoU step re-reads exactly that)"""
    box, _, _ = device_result
    sub = box[:2_000_000]
    pts = sub.reshape(-1, 2)
    off = np.arange(0, len(pts) + 1, 2, dtype=np.int32)
    again, arg = native.bbox_minmax(pts, off)
    assert np.array_equal(again.view(np.uint64), sub.view(np.uint64))
    assert (arg[:, :2] == 0).all()


def test_iou_flag_properties(native, table, device_result):
    box, _, high = device_result
    off = table.box_off
    n = np.diff(off)
    assert not high[n < 2].any()                                  # a row with one box has no pair
    lo = native.iou_any_ge(box, off, 2, 0.5)
    hi = native.iou_any_ge(box, off, 2, 0.999)
    assert ((hi <= high) & (high <= lo)).all()                    # monotone in the threshold
    assert np.array_equal(native.iou_any_ge(box, off, 2, 0.0), (n >= 2).astype(np.uint8))   # thr <= 0: any pair hits
    assert not native.iou_any_ge(box, off, 33, 0.0).any()         # min_boxes above every row
    flags, mx = native.iou_any_ge(box, off, 2, 0.98, want_max=True)
    assert np.array_equal(flags, high)
    assert np.array_equal(high.astype(bool), mx >= 0.98)          # flag == (max pair IoU >= thr)
    # reversing the boxes inside every row cannot change an any-pair predicate
    rev = np.concatenate([box[s:e][::-1] for s, e in zip(off[:2001], off[1:2001])])
    assert np.array_equal(native.iou_any_ge(rev, off[:2001] - off[0], 2, 0.98), high[:2000])


def test_dedup_properties(native):
    rng = np.random.default_rng(9)
    n = 4_000_000
    ids = rng.integers(0, int(0.9 * n), size=n)
    h = np.empty((n, 2), np.uint64)
    h[:, 0] = ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    h[:, 1] = ids.astype(np.uint64) ^ np.uint64(0xABCDEF)
    first, last, none = (native.dedup(h, k).astype(bool) for k in ("first", "last", False))
    uniq, idx_first, counts = np.unique(ids, return_index=True, return_counts=True)
    assert first.sum() == len(uniq) == last.sum()
    assert np.array_equal(np.flatnonzero(first), np.sort(idx_first))
    assert none.sum() == (counts == 1).sum() and not (none & ~first).any() and not (none & ~last).any()
    assert np.array_equal(native.dedup(h[first], "first"), np.ones(first.sum(), np.uint8))   # idempotent
    assert native.isin(h, h[first]).all() and not native.isin(h[first], np.zeros((0, 2), np.uint64)).any()


def test_split_properties(native):
    rng = np.random.default_rng(4)
    n, n_cat = 3_000_000, 7
    cat = rng.integers(-1, n_cat, size=n).astype(np.int32)
    sizes = np.bincount(cat[cat >= 0], minlength=n_cat).astype(np.int64)
    off = np.zeros(n_cat + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    perm = np.concatenate([native.mt19937_permutation(42, int(s)) for s in sizes])
    tr, va = (sizes * 0.8).astype(np.int64), (sizes * 0.1).astype(np.int64)
    split, pos = native.split_ids(cat, perm, off, tr, va)
    assert (split[cat < 0] == 255).all() and (pos[cat < 0] == -1).all()
    for c in range(n_cat):
        m = cat == c
        assert np.array_equal(np.sort(pos[m]), np.arange(sizes[c]))                 # a permutation of the category
        assert (split[m] == 0).sum() == tr[c] and (split[m] == 1).sum() == va[c]
        rank = np.arange(sizes[c])
        assert np.array_equal(perm[off[c]:off[c + 1]][pos[m]], rank)                # perm[pos] == in-category rank
