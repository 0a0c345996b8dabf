"""K2's sort-and-sweep path for rows of 96..256 boxes (csrc/k2_sweep.h) against the oracle's double loop
(reference core/processor.py:328-339, :368-376): the window filter may only admit pairs, never lose one."""
import numpy as np
import pytest

from oracle import lib as olib

pytestmark = pytest.mark.gpu

THR = [(0.98, 2), (0.5, 3), (0.05, 2), (1.0, 2), (1e-12, 2), (1.5, 2), (0.0, 2), (-1.0, 2), (0.98, 300)]


def _offsets(sizes):
    off = np.zeros(len(sizes) + 1, np.int32)
    np.cumsum(sizes, out=off[1:])
    return off


def _plant(rng, box, off, every=1):
    """near-duplicate, boundary, identical and swapped-corner partners somewhere inside every `every`-th row"""
    for r in range(0, len(off) - 1, every):
        s, e = int(off[r]), int(off[r + 1])
        if e - s < 2:
            continue
        i, j = rng.choice(np.arange(s, e), size=2, replace=False)
        mode = (r // every) % 5
        box[j] = box[i]
        if mode == 1:
            box[j, 3] -= (box[j, 3] - box[j, 1]) * 0.02              # at the 0.98 boundary
        elif mode == 2:
            box[j, 0] += (box[j, 2] - box[j, 0]) * 0.0201            # just outside it, shifted in x (the sweep axis)
        elif mode == 3:
            box[j] = box[j][[2, 3, 0, 1]]                            # un-normalised corners
        elif mode == 4:
            box[j, 0] += (box[j, 2] - box[j, 0]) * 0.0199            # just inside, shifted in x
    return box


def _tables():
    rng = np.random.default_rng(20260301)
    out = {}
    sizes = [256, 96, 95, 100, 128, 129, 200, 255, 256, 7, 0, 1, 256, 130, 97, 64, 250, 33, 40, 48, 63, 65, 70] * 3
    off = _offsets(sizes)
    B = int(off[-1])
    c = rng.random((B, 2)) * [1920, 1080]
    wh = rng.random((B, 2)) * 100 + 1
    out["uniform"] = (_plant(rng, np.concatenate([c, c + wh], axis=1), off), off)
    out["integers"] = (_plant(rng, np.round(np.concatenate([c, c + wh], axis=1), 0), off), off)
    # columns: many boxes share x1 exactly (the whole window is inside one key bucket), stacked in y
    col = np.repeat(rng.integers(0, 4, size=B) * 300.0, 1)
    yy = rng.random(B) * 5000
    out["columns"] = (_plant(rng, np.stack([col, yy, col + 200, yy + 30], axis=1), off), off)
    # ONE column: the x1 order cannot spread the row at all — the sweep gives up after n trips and sorts along the diagonal instead
    yy1 = rng.random(B) * 20000
    out["one_column"] = (_plant(rng, np.stack([np.full(B, 100.0), yy1, np.full(B, 300.0), yy1 + 30], axis=1), off), off)
    # everything on one spot: the sweep degenerates to all pairs
    jit = rng.random((B, 4)) * 1e-3
    out["one_spot"] = (_plant(rng, np.array([100.0, 100.0, 180.0, 160.0]) + jit, off, every=3), off)
    # magnitudes where f32 cannot tell the boxes apart (2^24 and beyond), negative and tiny coordinates
    big = np.concatenate([c, c + wh], axis=1) * 4096.0 + 16777216.123
    out["beyond_f32"] = (_plant(rng, big, off), off)
    out["negative"] = (_plant(rng, np.concatenate([c, c + wh], axis=1) - [2000, 1000, 2000, 1000], off), off)
    out["tiny"] = (_plant(rng, np.concatenate([c, c + wh], axis=1) * 1e-300, off), off)
    out["huge"] = (_plant(rng, np.concatenate([c, c + wh], axis=1) * 1e300, off), off)      # w*h overflows: inf areas
    out["small"] = (_plant(rng, np.concatenate([c, c + wh], axis=1) * 1e-150, off), off)    # every f32 key is 0
    out["large"] = (_plant(rng, np.concatenate([c, c + wh], axis=1) * 1e150, off), off)     # every f32 key is FLT_MAX
    # not finite somewhere in a third of the rows: those rows take the all-pairs code
    odd = _plant(rng, np.concatenate([c, c + wh], axis=1), off)
    for r in range(0, len(sizes), 3):
        if sizes[r] >= 2:
            k = rng.integers(off[r], off[r + 1])
            odd[k, rng.integers(0, 4)] = [np.nan, np.inf, -np.inf][r % 3]
    out["not_finite"] = (odd, off)
    # zero-width and zero-height boxes, nested boxes
    z = np.concatenate([c, c + wh], axis=1)
    z[::7, 2] = z[::7, 0]
    z[3::11, 3] = z[3::11, 1]
    z[5::13] = z[4::13][:len(z[5::13])] + [1, 1, -1, -1]
    out["degenerate"] = (_plant(rng, z, off), off)
    return out


TABLES = _tables()


@pytest.mark.parametrize("variant", [-1, 3, 5, 2])
@pytest.mark.parametrize("name", sorted(TABLES))
def test_sweep_flags_and_maximum(native, name, variant):
    box, off = TABLES[name]
    L = native.lib()
    native.check(L.dyd_set_option(b"k2_variant", variant), "opt")
    try:
        got = {(thr, mb): native.iou_any_ge(box, off, mb, thr) for thr, mb in THR}
        gmx = native.iou_any_ge(box, off, 2, 0.7, want_max=True)
    finally:
        native.check(L.dyd_set_option(b"k2_variant", -1), "opt")
    for (thr, mb), g in got.items():
        want = olib.iou_any_ge(box, off, mb, thr)
        assert np.array_equal(g, want), (name, thr, mb, np.flatnonzero(g != want)[:5])
    wmx = olib.iou_any_ge(box, off, 2, 0.7, want_max=True)
    assert np.array_equal(gmx[0], wmx[0])
    assert np.array_equal(gmx[1].view(np.uint64), wmx[1].view(np.uint64)), name


def test_sweep_tables_do_hit(native):
    """sanity of the tables themselves: the planted partners make some rows HIGH and leave others not"""
    for name in ("uniform", "columns", "one_column", "beyond_f32", "small", "large"):
        box, off = TABLES[name]
        want = olib.iou_any_ge(box, off, 2, 0.98)
        assert 0 < want.sum() < len(want), name


@pytest.mark.parametrize("variant", [-1, 6, 9, 10])
@pytest.mark.parametrize("bpr", [40, 48, 64, 65, 100, 128, 129, 200, 256])
def test_fused_dense_rows(native, bpr, variant):
    """the fused launch on tables of dense rows: workgroup tiles (6, 9) and the wave kernel's dense instantiation (10) hand rows of
    40..256 boxes to the sweep"""
    from deal_yolo_daya_amd import synth
    t = synth.generate(300, seed=bpr, boxes_per_row=bpr)
    L = native.lib()
    native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
    try:
        res = {thr: native.bbox_iou_fused(t.xy, t.pt_off, t.box_off, 2, thr, want_box=True) for thr in (0.98, 0.3)}
    finally:
        native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
    for thr, (arg, high, box) in res.items():
        obox, oarg, ohigh = olib.bbox_iou_chain(t.xy, t.pt_off, t.box_off, 2, thr)
        assert np.array_equal(arg, oarg) and np.array_equal(box.view(np.uint64), obox.view(np.uint64))
        assert np.array_equal(high, ohigh), (bpr, variant, thr)
        assert 0 < ohigh.sum()


@pytest.mark.parametrize("variant", [10, -1])
def test_fused_dense_mixed_rows_and_empty_polygons(native, variant):
    """rows of every size between 1 and 300 boxes in one table, some polygons without a valid point (the row's IoU list ends there,
    reference processor.py:254-255 -> :364-365), NaN and inf points: flags, boxes and arg indices against the chain oracle"""
    rng = np.random.default_rng(77)
    sizes = np.concatenate([rng.integers(1, 301, size=400), [256, 255, 257, 64, 65, 40, 39, 128, 129, 300]])
    box_off = np.zeros(len(sizes) + 1, np.int32)
    np.cumsum(sizes, out=box_off[1:])
    B = int(box_off[-1])
    npts = rng.integers(1, 9, size=B)
    npts[rng.integers(0, B, size=120)] = 0                 # polygons without a valid point
    pt_off = np.zeros(B + 1, np.int32)
    np.cumsum(npts, out=pt_off[1:])
    P = int(pt_off[-1])
    centre = rng.random((B, 2)) * [1920, 1080]
    xy = np.repeat(centre, npts, axis=0) + rng.random((P, 2)) * 100 - 50
    xy = np.round(xy, 1)
    for r in range(0, len(sizes), 3):                      # a near-duplicate of a random box at the end of every third row
        s, e = int(box_off[r]), int(box_off[r + 1])
        if e - s >= 2:
            src, dst = int(rng.integers(s, e - 1)), e - 1
            k = min(int(npts[src]), int(npts[dst]))
            if k == 0:
                continue
            xy[pt_off[dst]:pt_off[dst] + k] = xy[pt_off[src]:pt_off[src] + k]
            xy[pt_off[dst] + k:pt_off[dst + 1]] = xy[pt_off[src]]
            if k < npts[src]:
                xy[pt_off[src] + k:pt_off[src + 1]] = xy[pt_off[src]]
    for b in rng.integers(0, B, size=60):                  # polygons of NaN points only
        xy[pt_off[b]:pt_off[b + 1]] = np.nan
    for b in rng.integers(0, B, size=60):                  # an inf or a NaN somewhere
        if npts[b]:
            xy[pt_off[b] + int(rng.integers(0, npts[b])), int(rng.integers(0, 2))] = [np.inf, -np.inf, np.nan][int(rng.integers(0, 3))]
    L = native.lib()
    native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
    try:
        res = {(thr, mb): native.bbox_iou_fused(xy, pt_off, box_off, mb, thr, want_box=True) for thr, mb in ((0.98, 2), (0.5, 3), (0.0, 2), (1.0, 2))}
    finally:
        native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
    for (thr, mb), (arg, high, box) in res.items():
        obox, oarg, ohigh = olib.bbox_iou_chain(xy, pt_off, box_off, mb, thr)
        assert np.array_equal(arg, oarg)
        assert np.array_equal(box.view(np.uint64), obox.view(np.uint64))
        assert np.array_equal(high, ohigh), (thr, mb, np.flatnonzero(high != ohigh)[:5], sizes[np.flatnonzero(high != ohigh)[:5]])
    assert 0 < res[(0.98, 2)][1].sum() < len(sizes)


@pytest.mark.parametrize("variant", [-1, 4])
@pytest.mark.parametrize("thr,mb", [(0.98, 2), (0.4, 3), (0.0, 2), (0.98, 100)])
def test_sparse_table_with_a_heavy_tail(native, variant, thr, mb):
    """a table of small rows (the sparse wave kernel's) with a few rows of 65..256 boxes and some beyond: the kernel queues them and
    the drain kernel sorts and sweeps them (k2_wave.h, k2_big_rows_kernel) — flags, boxes and arg indices against the chain oracle,
    empty polygons (the list's end) and non-finite corners inside the long rows included; K2 alone on the same boxes as well"""
    rng = np.random.default_rng(5)
    sizes = rng.integers(0, 21, size=4000)
    sizes[rng.integers(0, 4000, size=90)] = rng.integers(65, 257, size=90)
    sizes[rng.integers(0, 4000, size=6)] = rng.integers(257, 700, size=6)
    sizes[[7, 8, 9]] = [65, 256, 64]
    box_off = np.zeros(len(sizes) + 1, np.int32)
    np.cumsum(sizes, out=box_off[1:])
    B = int(box_off[-1])
    assert B <= 32 * len(sizes)                               # the automatic choice is the sparse kernel
    npts = rng.integers(1, 7, size=B)
    npts[rng.integers(0, B, size=40)] = 0
    pt_off = np.zeros(B + 1, np.int32)
    np.cumsum(npts, out=pt_off[1:])
    centre = rng.random((B, 2)) * [1920, 1080]
    xy = np.round(np.repeat(centre, npts, axis=0) + rng.random((int(pt_off[-1]), 2)) * 80 - 40, 1)
    for r in np.flatnonzero(sizes >= 2)[::2]:                 # a copy of a random box at the end of every other row
        s0, e0 = int(box_off[r]), int(box_off[r + 1])
        src, dst = int(rng.integers(s0, e0 - 1)), e0 - 1
        k = min(int(npts[src]), int(npts[dst]))
        if k:
            xy[pt_off[dst]:pt_off[dst] + k] = xy[pt_off[src]:pt_off[src] + k]
            xy[pt_off[dst] + k:pt_off[dst + 1]] = xy[pt_off[src]]
            xy[pt_off[src] + k:pt_off[src + 1]] = xy[pt_off[src]]
    xy[rng.integers(0, len(xy), 12), rng.integers(0, 2, 12)] = rng.choice([np.nan, np.inf, -np.inf], 12)
    L = native.lib()
    native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
    try:
        arg, high, box = native.bbox_iou_fused(xy, pt_off, box_off, mb, thr, want_box=True)
    finally:
        native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
    obox, oarg, ohigh = olib.bbox_iou_chain(xy, pt_off, box_off, mb, thr)
    assert np.array_equal(arg, oarg) and np.array_equal(box.view(np.uint64), obox.view(np.uint64))
    bad = np.flatnonzero(high != ohigh)
    assert len(bad) == 0, (bad[:6].tolist(), sizes[bad[:6]].tolist())
    if thr == 0.98 and mb == 2:
        assert ohigh[sizes > 64].sum() > 10
    kbox = np.where(np.isnan(obox), 0.0, obox)
    native.check(L.dyd_set_option(b"k2_variant", 4 if variant == 4 else -1), "opt")
    try:
        got = native.iou_any_ge(kbox, box_off, mb, thr)
    finally:
        native.check(L.dyd_set_option(b"k2_variant", -1), "opt")
    assert np.array_equal(got, olib.iou_any_ge(kbox, box_off, mb, thr))


@pytest.mark.parametrize("thr", [float("nan"), float("inf"), -float("inf"), 1e-300, 5e-324, 1.0000000000000002, 0.9999999999999999])
def test_odd_thresholds_on_dense_rows(native, thr):
    """thresholds no UI sends but the C ABI accepts: the sweep's window arithmetic must not turn them into something else (NaN: nothing
    is >= NaN; inf: nothing; tiny positive: every intersecting pair; the neighbours of 1.0: identical boxes only / also almost identical)"""
    box, off = TABLES["integers"]
    L = native.lib()
    for variant in (-1, 3, 5):
        native.check(L.dyd_set_option(b"k2_variant", variant), "opt")
        try:
            got = native.iou_any_ge(box, off, 2, thr)
            gmx = native.iou_any_ge(box, off, 2, thr, want_max=True)
        finally:
            native.check(L.dyd_set_option(b"k2_variant", -1), "opt")
        want = olib.iou_any_ge(box, off, 2, thr, want_max=True)
        assert np.array_equal(got, want[0]) and np.array_equal(gmx[0], want[0]), (thr, variant)
        assert np.array_equal(gmx[1].view(np.uint64), want[1].view(np.uint64)), (thr, variant)
    t = TABLES["uniform"]
    from deal_yolo_daya_amd import synth
    tb = synth.generate(200, seed=3, boxes_per_row=150)
    for variant in (-1, 4, 6, 10):
        native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
        try:
            arg, high = native.bbox_iou_fused(tb.xy, tb.pt_off, tb.box_off, 2, thr)
        finally:
            native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
        _, oarg, ohigh = olib.bbox_iou_chain(tb.xy, tb.pt_off, tb.box_off, 2, thr)
        assert np.array_equal(arg, oarg) and np.array_equal(high, ohigh), (thr, variant)


@pytest.mark.parametrize("variant", [-1, 4, 6, 9, 10])
def test_fused_rows_in_one_column(native, variant):
    """rows whose boxes all share x1 (text lines, table cells): the worst case of the x1 order (every box in every other's window) and, in the
    drain kernel, the rows its trip budget sends to the diagonal order — same flags as the chain oracle through every fused variant"""
    rng = np.random.default_rng(9)
    sizes = np.array([256, 200, 130, 100, 64, 50, 41, 300, 700, 20, 256, 256] * 4)
    box_off = _offsets(sizes)
    B = int(box_off[-1])
    y = rng.random(B) * 30000
    x0 = np.repeat(rng.integers(0, 3, size=len(sizes)) * 500.0, sizes)
    corners = np.stack([x0, y, x0 + 200, y + 30], axis=1)
    for r in range(0, len(sizes), 2):                          # a near copy somewhere in every other row
        s0, e0 = int(box_off[r]), int(box_off[r + 1])
        i, j = rng.choice(np.arange(s0, e0), size=2, replace=False)
        corners[j] = corners[i]
        corners[j, 3] -= 0.3 * ((r // 2) % 3)                   # identical, 1 % lower, 2 % lower
    xy = corners[:, [0, 1, 2, 1, 2, 3, 0, 3]].reshape(-1, 2)    # four points per polygon
    pt_off = (np.arange(B + 1) * 4).astype(np.int32)
    L = native.lib()
    native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
    try:
        res = {thr: native.bbox_iou_fused(xy, pt_off, box_off, 2, thr) for thr in (0.98, 0.5)}
    finally:
        native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
    for thr, (arg, high) in res.items():
        _, oarg, ohigh = olib.bbox_iou_chain(xy, pt_off, box_off, 2, thr)
        assert np.array_equal(arg, oarg) and np.array_equal(high, ohigh), (variant, thr, np.flatnonzero(high != ohigh)[:5].tolist())
        assert 0 < ohigh.sum() < len(sizes)
