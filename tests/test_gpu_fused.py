"""The fused K1+K2 launch (dyd_bbox_iou_fused_dev) and the device-pointer (_dev) entry points on
HBM-resident tensors, against the CPU oracle.  Needs a real MI355X (-m gpu)."""
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
@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3, 4])
def test_fused_matches_oracle(native, n_rows, max_boxes, max_pts, special, variant):
    import torch

    rng = np.random.default_rng(n_rows * 13 + max_boxes)
    xy, pt_off, box_off = _table(rng, n_rows, max_boxes, max_pts, special)
    B = len(pt_off) - 1
    obox, oarg = olib.bbox_minmax(xy, pt_off)
    L = native.lib()
    dev = torch.device("cuda:0")
    t_xy = torch.from_numpy(xy).to(dev) if len(xy) else torch.zeros((1, 2), dtype=torch.float64, device=dev)
    t_po, t_bo = torch.from_numpy(pt_off).to(dev), torch.from_numpy(box_off).to(dev)
    for thr, mb in ((0.98, 2), (0.5, 3)):
        ohigh = olib.iou_any_ge(obox, box_off, mb, thr)
        t_box = torch.full((max(B, 1), 4), -7.0, dtype=torch.float64, device=dev)
        t_arg = torch.full((max(B, 1), 4), -7, dtype=torch.int32, device=dev)
        t_high = torch.full((n_rows,), 9, dtype=torch.uint8, device=dev)
        native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
        try:
            native.check(L.dyd_bbox_iou_fused_dev(t_xy.data_ptr(), t_po.data_ptr(), t_bo.data_ptr(), n_rows, B, mb, thr,
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
        native.check(L.dyd_bbox_minmax_dev(t_xy.data_ptr(), t_po.data_ptr(), B, t_box.data_ptr(), t_arg.data_ptr(),
                                           s.cuda_stream), "k1")
        native.check(L.dyd_iou_any_ge_dev(t_box.data_ptr(), t_bo.data_ptr(), 500, 2, 0.9, t_high.data_ptr(), None,
                                          s.cuda_stream), "k2")
    s.synchronize()
    obox, oarg = olib.bbox_minmax(xy, pt_off)
    assert np.array_equal(t_arg.cpu().numpy(), oarg)
    assert np.array_equal(t_high.cpu().numpy(), olib.iou_any_ge(obox, box_off, 2, 0.9))


@pytest.mark.parametrize("n_rows,max_boxes,fixed", [(500, 32, None), (40, None, 256), (3, None, 1500)])
def test_k2_small_tile_variant(native, n_rows, max_boxes, fixed):
    """K2 with 8-row / 128-box wave tiles (rows above 128 boxes take the streaming path)."""
    from helpers import random_boxes
    rng = np.random.default_rng(n_rows)
    box, off = random_boxes(rng, n_rows, max_boxes or 1, fixed=fixed)
    L = native.lib()
    native.check(L.dyd_set_option(b"k2_variant", 1), "opt")
    try:
        got = native.iou_any_ge(box, off, 2, 0.98)
        gmx = native.iou_any_ge(box, off, 2, 0.5, want_max=True)
    finally:
        native.check(L.dyd_set_option(b"k2_variant", 0), "opt")
    assert np.array_equal(got, olib.iou_any_ge(box, off, 2, 0.98))
    wmx = olib.iou_any_ge(box, off, 2, 0.5, want_max=True)
    assert np.array_equal(gmx[0], wmx[0]) and np.array_equal(gmx[1].view(np.uint64), wmx[1].view(np.uint64))
