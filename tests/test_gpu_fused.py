"""The fused K1+K2 launch (dyd_bbox_iou_fused_dev / dyd_bbox_iou_fused) and the device-pointer (_dev) entry points on
HBM-resident tensors, against the CPU oracle of the reference's replace -> IoU chain (oracle.lib.bbox_iou_chain: a row's IoU box
list ends at its first polygon without a valid point, processor.py:254-255 -> :364-365).  Needs a real MI355X (-m gpu)."""
import numpy as np
import pytest

from helpers import random_polygons
from oracle import lib as olib

pytestmark = pytest.mark.gpu


def _table(rng, n_rows, max_boxes, max_pts, special):
    nb = rng.integers(0, max_boxes + 1, size=n_rows)
    box_off = np.zeros(n_rows + 1, np.int32)
    np.cumsum(nb, out=box_off[1:])
    xy, pt_off = random_polygons(rng, int(box_off[-1]), max_pts, special=special)
    # plant near-duplicate polygons so that some rows are HIGH
    for r in rng.integers(0, n_rows, size=max(1, n_rows // 3)):
        s, e = box_off[r], box_off[r + 1]
        if e - s >= 2:
            a, b = pt_off[s], pt_off[s + 1]
            c, d = pt_off[e - 1], pt_off[e]
            k = min(b - a, d - c)
            if k:
                xy[c:c + k] = xy[a:a + k]
    return xy, pt_off, box_off


@pytest.mark.parametrize("n_rows,max_boxes,max_pts,special", [(1, 3, 5, False), (63, 32, 12, True), (64, 32, 12, True),
                                                              (65, 32, 12, True), (3000, 32, 12, True),
                                                              (700, 90, 30, False), (50, 300, 6, False),
                                                              (20, 4, 900, True)])
@pytest.mark.parametrize("variant", [-1, 1, 4, 6, 9, 10])
def test_fused_matches_oracle(native, n_rows, max_boxes, max_pts, special, variant):
    import torch

    rng = np.random.default_rng(n_rows * 13 + max_boxes)
    xy, pt_off, box_off = _table(rng, n_rows, max_boxes, max_pts, special)
    B = len(pt_off) - 1
    L = native.lib()
    dev = torch.device("cuda:0")
    t_xy = torch.from_numpy(xy).to(dev) if len(xy) else torch.zeros((1, 2), dtype=torch.float64, device=dev)
    t_po, t_bo = torch.from_numpy(pt_off).to(dev), torch.from_numpy(box_off).to(dev)
    for thr, mb in ((0.98, 2), (0.5, 3)):
        obox, oarg, ohigh = olib.bbox_iou_chain(xy, pt_off, box_off, mb, thr)
        t_box = torch.full((max(B, 1), 4), -7.0, dtype=torch.float64, device=dev)
        t_arg = torch.full((max(B, 1), 4), -7, dtype=torch.int32, device=dev)
        t_high = torch.full((n_rows,), 9, dtype=torch.uint8, device=dev)
        native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
        try:
            native.check(L.dyd_bbox_iou_fused_dev(t_xy.data_ptr(), t_po.data_ptr(), t_bo.data_ptr(), n_rows, B, int(t_xy.shape[0]), mb, thr,
                                                  t_box.data_ptr(), t_arg.data_ptr(), t_high.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "fused")
        finally:
            native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
        torch.cuda.synchronize()
        box, arg, high = t_box.cpu().numpy()[:B], t_arg.cpu().numpy()[:B], t_high.cpu().numpy()
        assert np.array_equal(arg, oarg)
        assert np.array_equal(np.isnan(box), np.isnan(obox))
        assert np.array_equal(box[~np.isnan(box)].view(np.uint64), obox[~np.isnan(obox)].view(np.uint64))
        assert np.array_equal(high, ohigh), (thr, mb)


def _chain_table(rng, n_rows, max_boxes, empty_share):
    """rows of disjoint boxes whose LAST box repeats the first, with empty polygons planted: "K2 on K1's boxes" calls such a row
    HIGH whenever its two ends are there, the reference chain only when no empty polygon comes before the last box"""
    nb = rng.integers(0, max_boxes + 1, size=n_rows)
    box_off = np.zeros(n_rows + 1, np.int32)
    np.cumsum(nb, out=box_off[1:])
    B = int(box_off[-1])
    npts = np.where(rng.random(B) < empty_share, 0, 4)
    pt_off = np.zeros(B + 1, np.int32)
    np.cumsum(npts, out=pt_off[1:])
    k_in_row = np.arange(B) - np.repeat(box_off[:-1], nb)
    k_in_row[(box_off[1:] - 1)[nb > 1]] = 0                      # the last box of a row sits on its first
    shift = np.repeat(k_in_row, npts)[:, None] * np.array([[150.0, 0.0]])
    corner = np.tile(np.array([[0.0, 0.0], [100.0, 0.0], [100.0, 100.0], [0.0, 100.0]]), (int(npts.sum()) // 4, 1))
    return corner + shift, pt_off, box_off


@pytest.mark.parametrize("n_rows,max_boxes,share", [(400, 6, 0.3), (3000, 32, 0.05), (300, 64, 0.02), (40, 200, 0.01), (6, 700, 0.003),
                                                    (50, 5, 1.0)])
@pytest.mark.parametrize("variant", [-1, 1, 4, 6, 9, 10])
def test_fused_ends_a_row_at_its_first_empty_polygon(native, n_rows, max_boxes, share, variant):
    """[A, null, A] is not HIGH, [A, A, null] is (reference processor.py:254-255 -> :364-365): every fused variant, rows below
    and above the 64 boxes of a wave tile, through the host-pointer entry"""
    rng = np.random.default_rng(n_rows + max_boxes)
    xy, pt_off, box_off = _chain_table(rng, n_rows, max_boxes, share)
    L = native.lib()
    native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
    try:
        got = {(mb, thr): native.bbox_iou_fused(xy, pt_off, box_off, mb, thr, want_box=True) for mb, thr in ((2, 0.98), (3, 0.5), (1, 0.0))}
    finally:
        native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
    plain_differs = False
    for (mb, thr), (arg, high, box) in got.items():
        obox, oarg, ohigh = olib.bbox_iou_chain(xy, pt_off, box_off, mb, thr)
        assert np.array_equal(arg, oarg)
        assert np.array_equal(np.isnan(box), np.isnan(obox)) and np.array_equal(box[~np.isnan(box)], obox[~np.isnan(obox)])
        assert np.array_equal(high, ohigh), (mb, thr, np.flatnonzero(high != ohigh)[:10])
        plain_differs |= not np.array_equal(ohigh, olib.iou_any_ge(obox, box_off, mb, thr))
    assert plain_differs or share == 1.0, "the table must tell the chain from K2-on-K1's-boxes"


@pytest.mark.parametrize("variant", [-1, 4, 6, 9, 10])
def test_rows_of_thousands_of_boxes_through_the_device_entries(native, variant):
    """a 5000-box row, a 700-box row with an empty polygon in its middle and one of 300 boxes among ordinary rows, device
    resident: the main kernel queues them and k2_big_rows_kernel spreads them over the grid (csrc/k2_wave.h) — same flags as
    the oracle's chain, for dyd_bbox_iou_fused_dev and for dyd_iou_any_ge_dev on the resulting boxes"""
    import torch

    rng = np.random.default_rng(17)
    nb = rng.integers(0, 20, size=400)
    nb[37], nb[200], nb[399], nb[5] = 5000, 700, 300, 257
    box_off = np.zeros(len(nb) + 1, np.int32)
    np.cumsum(nb, out=box_off[1:])
    B = int(box_off[-1])
    npts = np.full(B, 4)
    npts[box_off[200] + 350] = 0                                   # the 700-box row ends at its 350th polygon
    pt_off = np.zeros(B + 1, np.int32)
    np.cumsum(npts, out=pt_off[1:])
    k_in_row = np.arange(B) - np.repeat(box_off[:-1], nb)
    corner = np.tile(np.array([[0.0, 0.0], [100.0, 0.0], [100.0, 100.0], [0.0, 100.0]]), (int(npts.sum()) // 4, 1))
    xy = corner + np.repeat(k_in_row, npts)[:, None] * np.array([[150.0, 0.0]])
    # the 5000-box row: its last box repeats box 4321 (the only HIGH pair lies deep inside); the 700-box row: boxes 699 and 10
    for r, a, b in ((37, 4999, 4321), (200, 699, 10), (399, 299, 298)):
        pa, pb = pt_off[box_off[r] + a], pt_off[box_off[r] + b]
        xy[pa:pa + 4] = xy[pb:pb + 4]
    L = native.lib()
    dev = torch.device("cuda:0")
    t_xy, t_po, t_bo = (torch.from_numpy(a).to(dev) for a in (xy, pt_off, box_off))
    n_rows = len(nb)
    native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
    try:
        for mb, thr in ((2, 0.98), (3, 0.5)):
            obox, oarg, ohigh = olib.bbox_iou_chain(xy, pt_off, box_off, mb, thr)
            t_box = torch.empty((B, 4), dtype=torch.float64, device=dev)
            t_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
            t_high = torch.full((n_rows,), 9, dtype=torch.uint8, device=dev)
            native.check(L.dyd_bbox_iou_fused_dev(t_xy.data_ptr(), t_po.data_ptr(), t_bo.data_ptr(), n_rows, B, len(xy), mb, thr,
                                                  t_box.data_ptr(), t_arg.data_ptr(), t_high.data_ptr(), None), "fused")
            torch.cuda.synchronize()
            assert np.array_equal(t_arg.cpu().numpy(), oarg)
            assert np.array_equal(t_high.cpu().numpy(), ohigh), (mb, thr, np.flatnonzero(t_high.cpu().numpy() != ohigh))
            assert ohigh[37] == 1 and ohigh[200] == 0 and ohigh[399] == 1
            # K2 alone on those boxes (no chain rule there: the 700-box row pairs all of its boxes; its empty polygon is a NaN box)
            t_high.fill_(9)
            native.check(L.dyd_iou_any_ge_dev(t_box.data_ptr(), t_bo.data_ptr(), n_rows, B, mb, thr, t_high.data_ptr(), None, None), "k2")
            torch.cuda.synchronize()
            assert np.array_equal(t_high.cpu().numpy(), olib.iou_any_ge(obox, box_off, mb, thr))
    finally:
        native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
    assert L.dyd_device_status(None) == 0


def test_fused_host_entry_edges(native):
    """no rows, rows without boxes, boxes without points"""
    arg, high = native.bbox_iou_fused(np.zeros((0, 2)), np.zeros(1, np.int32), np.zeros(1, np.int32), 2, 0.98)
    assert arg.shape == (0, 4) and high.shape == (0,)
    arg, high = native.bbox_iou_fused(np.zeros((0, 2)), np.zeros(1, np.int32), np.zeros(4, np.int32), 2, 0.98)
    assert arg.shape == (0, 4) and high.tolist() == [0, 0, 0]
    arg, high = native.bbox_iou_fused(np.zeros((0, 2)), np.zeros(4, np.int32), np.array([0, 3], np.int32), 2, 0.0)
    assert (arg == -1).all() and high.tolist() == [0]
    with pytest.raises(ValueError):
        native.bbox_iou_fused(np.zeros((1, 2)), np.array([0, 2], np.int32), np.array([0, 1], np.int32), 2, 0.98)


def test_dev_entry_points_on_a_side_stream(native):
    """_dev twins launch on exactly the stream they are given (here a non-default torch stream)."""
    import torch

    rng = np.random.default_rng(2)
    xy, pt_off, box_off = _table(rng, 500, 20, 10, False)
    B = len(pt_off) - 1
    dev = torch.device("cuda:0")
    L = native.lib()
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        t_xy, t_po, t_bo = (torch.from_numpy(a).to(dev) for a in (xy, pt_off, box_off))
        t_box = torch.empty((B, 4), dtype=torch.float64, device=dev)
        t_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
        t_high = torch.empty(500, dtype=torch.uint8, device=dev)
        native.check(L.dyd_bbox_minmax_dev(t_xy.data_ptr(), t_po.data_ptr(), B, int(t_xy.shape[0]), t_box.data_ptr(), t_arg.data_ptr(),
                                           s.cuda_stream), "k1")
        native.check(L.dyd_iou_any_ge_dev(t_box.data_ptr(), t_bo.data_ptr(), 500, B, 2, 0.9, t_high.data_ptr(), None,
                                          s.cuda_stream), "k2")
    s.synchronize()
    obox, oarg = olib.bbox_minmax(xy, pt_off)
    assert np.array_equal(t_arg.cpu().numpy(), oarg)
    assert np.array_equal(t_high.cpu().numpy(), olib.iou_any_ge(obox, box_off, 2, 0.9))


@pytest.mark.parametrize("n_rows,max_boxes,fixed", [(500, 32, None), (2000, 60, None), (40, None, 256), (3, None, 1500),
                                                    (2, None, 700)])
@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3, 4])
@pytest.mark.parametrize("special", [True, False])
def test_k2_variants(native, n_rows, max_boxes, fixed, variant, special):
    """K2 tile variants: 1 = 8-row / 128-box wave tiles, 2 / 3 = the f32 reject filter in front of the
    exact test (16 / 8-row tiles), 4 = the fused wave kernel's pair stage alone, -1 = by the table's shape.  Rows above
    the tile capacity take the streaming path."""
    from helpers import random_boxes
    rng = np.random.default_rng(n_rows + (fixed or 0))
    box, off = random_boxes(rng, n_rows, max_boxes or 1, fixed=fixed, special=special)
    if not special:      # big coordinates: the outward f32 rounding matters (f32 has 24 bits)
        box = box * 4096.0 + 0.123456789
    L = native.lib()
    native.check(L.dyd_set_option(b"k2_variant", variant), "opt")
    try:
        res = {(mb, thr): native.iou_any_ge(box, off, mb, thr) for mb, thr in ((2, 0.98), (3, 0.5), (2, 0.0), (2, 1.0))}
        gmx = native.iou_any_ge(box, off, 2, 0.5, want_max=True)
    finally:
        native.check(L.dyd_set_option(b"k2_variant", -1), "opt")
    for (mb, thr), got in res.items():
        assert np.array_equal(got, olib.iou_any_ge(box, off, mb, thr)), (mb, thr)
    wmx = olib.iou_any_ge(box, off, 2, 0.5, want_max=True)
    assert np.array_equal(gmx[0], wmx[0]) and np.array_equal(gmx[1].view(np.uint64), wmx[1].view(np.uint64))


def test_k2_filter_is_conservative_at_f32_resolution(native):
    """boxes that overlap by less than one f32 ulp: the filter must keep them and the exact test decide"""
    L = native.lib()
    base = 16777216.0                                     # 2^24: f32 spacing is 2 here, f64 sees 1e-9
    rows = []
    for eps in (1e-9, 1e-6, 0.5, 1.0, 2.0, -1e-9, 0.0):
        a = [base, base, base + 10.0, base + 10.0]
        b = [base + 10.0 - eps, base, base + 20.0, base + 10.0]       # overlaps a by `eps` in x
        c = list(a)                                                    # exact duplicate of a -> IoU 1
        rows.append(np.array([a, b], np.float64))
        rows.append(np.array([a, b, c], np.float64))
    box = np.concatenate(rows)
    off = np.zeros(len(rows) + 1, np.int32)
    np.cumsum([len(r) for r in rows], out=off[1:])
    for variant in (0, 2, 3, 4, -1):
        native.check(L.dyd_set_option(b"k2_variant", variant), "opt")
        try:
            for thr in (1e-12, 0.5, 0.98):
                assert np.array_equal(native.iou_any_ge(box, off, 2, thr), olib.iou_any_ge(box, off, 2, thr)), (variant, thr)
            g = native.iou_any_ge(box, off, 2, 0.5, want_max=True)[1]
        finally:
            native.check(L.dyd_set_option(b"k2_variant", -1), "opt")
        assert np.array_equal(g.view(np.uint64), olib.iou_any_ge(box, off, 2, 0.5, want_max=True)[1].view(np.uint64))
