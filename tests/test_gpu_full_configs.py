"""BASELINE configs[2] and configs[4] at FULL size on one MI355X, checked through size-independent properties
and exact oracle runs on samples (a full oracle run would take tens of minutes at these sizes):

  configs[2]  10 M rows: K3 -> K4 -> K5 -> fused K1+K2 -> K6, every input resident in HBM
  configs[4]  10 M rows x 256 boxes/image (2.56 G boxes, 82 GB): pairwise IoU in chunks of 5 M rows, because the
              C ABI's row offsets are int32 and 2.56 G boxes do not fit one offset array

Inputs are generated on the device with torch.  Needs a real MI355X (-m gpu).
"""
import numpy as np
import pytest

from oracle import lib as olib

pytestmark = pytest.mark.gpu


def _dev():
    import torch
    return torch, torch.device("cuda:0")


def test_config2_ten_million_rows_pipeline(native):
    torch, dev = _dev()
    L, ck = native.lib(), native.check
    sp = torch.cuda.current_stream().cuda_stream
    N = 10_000_000
    g = torch.Generator(device=dev).manual_seed(2026)
    # ---- K3 + K4 + K5 on fixed-width keys: "http://img.example/<9 digits>.jpg" -------------------------------
    ids = torch.randint(0, int(0.9 * N) + 1, (N,), generator=g, device=dev, dtype=torch.int64)
    prefix, suffix = b"http://img.example/", b".jpg"
    width = len(prefix) + 9 + len(suffix)

    def url_bytes(idv):
        n = idv.numel()
        out = torch.empty((n, width), dtype=torch.uint8, device=dev)
        out[:, :len(prefix)] = torch.tensor(list(prefix), dtype=torch.uint8, device=dev)
        out[:, len(prefix) + 9:] = torch.tensor(list(suffix), dtype=torch.uint8, device=dev)
        v = idv.clone()
        for k in range(8, -1, -1):
            out[:, len(prefix) + k] = (v % 10 + 48).to(torch.uint8)
            v //= 10
        return out.reshape(-1), torch.arange(n + 1, device=dev, dtype=torch.int64) * width

    data, off = url_bytes(ids)
    h = torch.empty((N, 2), dtype=torch.int64, device=dev)
    ck(L.dyd_hash128_dev(data.data_ptr(), off.data_ptr(), N, h.data_ptr(), sp), "k3")
    sample = torch.randint(0, N, (2000,), generator=torch.Generator().manual_seed(1)).tolist()
    hs = h.cpu().numpy().view(np.uint64)
    host_data, host_off = data.cpu().numpy(), off.cpu().numpy()
    want = olib.hash128(np.concatenate([host_data[host_off[i]:host_off[i + 1]] for i in sample]),
                        np.arange(len(sample) + 1, dtype=np.int64) * width)
    assert np.array_equal(hs[sample], want)                                          # hashes: exact on a sample
    keep = {}
    for mode, name in ((0, "first"), (1, "last"), (2, "none")):
        k = torch.empty(N, dtype=torch.uint8, device=dev)
        ck(L.dyd_dedup_dev(h.data_ptr(), N, mode, k.data_ptr(), sp), "k4")
        keep[name] = k.bool()
    uniq, inv, counts = torch.unique(ids, return_inverse=True, return_counts=True)
    first_idx = torch.full((uniq.numel(),), N, dtype=torch.int64, device=dev).scatter_reduce(0, inv, torch.arange(N, device=dev), "amin")
    last_idx = torch.full((uniq.numel(),), -1, dtype=torch.int64, device=dev).scatter_reduce(0, inv, torch.arange(N, device=dev), "amax")
    assert int(keep["first"].sum()) == uniq.numel() == int(keep["last"].sum())
    assert torch.equal(torch.nonzero(keep["first"]).flatten(), torch.sort(first_idx).values)
    assert torch.equal(torch.nonzero(keep["last"]).flatten(), torch.sort(last_idx).values)
    assert torch.equal(keep["none"], counts[inv] == 1)
    ref_ids = torch.arange(0, int(0.9 * N) + 1, 10, device=dev, dtype=torch.int64)
    rdata, roff = url_bytes(ref_ids)
    R = ref_ids.numel()
    hr = torch.empty((R, 2), dtype=torch.int64, device=dev)
    ck(L.dyd_hash128_dev(rdata.data_ptr(), roff.data_ptr(), R, hr.data_ptr(), sp), "k3r")
    hit = torch.empty(N, dtype=torch.uint8, device=dev)
    ck(L.dyd_isin_dev(h.data_ptr(), N, hr.data_ptr(), R, hit.data_ptr(), sp), "k5")
    assert torch.equal(hit.bool(), ids % 10 == 0)                                    # membership: exact, every row
    del data, off, h, hr, hit, rdata, roff, keep, uniq, inv, counts, first_idx, last_idx

    # ---- fused K1 + K2: 10 M rows, 1..32 boxes of 3..12 points ---------------------------------------------------
    nbox = torch.randint(1, 33, (N,), generator=g, device=dev, dtype=torch.int32)
    box_off = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    box_off[1:] = torch.cumsum(nbox, 0, dtype=torch.int64).to(torch.int32)
    B = int(box_off[-1])
    npts = torch.randint(3, 13, (B,), generator=g, device=dev, dtype=torch.int32)
    pt_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    pt_off[1:] = torch.cumsum(npts, 0, dtype=torch.int64).to(torch.int32)
    P = int(pt_off[-1])
    centre = torch.rand((B, 2), generator=g, device=dev, dtype=torch.float64) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)
    xy = torch.repeat_interleave(centre, npts.long(), dim=0) + torch.rand((P, 2), generator=g, device=dev, dtype=torch.float64) * 100 - 50
    xy = torch.round(xy * 100) / 100
    del centre
    # the last box of every 20th row copies the row's first polygon, 1 % flatter: IoU ~ 0.99 -> HIGH
    rows_dup = torch.arange(0, N, 20, device=dev)
    rows_dup = rows_dup[nbox[rows_dup] >= 2]
    b_first, b_last = box_off[rows_dup].long(), box_off[rows_dup + 1].long() - 1
    same = npts[b_first] == npts[b_last]
    b_first, b_last, rows_dup = b_first[same], b_last[same], rows_dup[same]
    cnt = npts[b_first].long()
    src = torch.repeat_interleave(pt_off[b_first].long(), cnt) + (torch.arange(int(cnt.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    dst = torch.repeat_interleave(pt_off[b_last].long(), cnt) + (src - torch.repeat_interleave(pt_off[b_first].long(), cnt))
    xy[dst] = xy[src]
    out_box = torch.empty((B, 4), dtype=torch.float64, device=dev)
    out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
    out_high = torch.empty(N, dtype=torch.uint8, device=dev)
    ck(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, int(xy.shape[0]), 2, 0.98, out_box.data_ptr(),
                                out_arg.data_ptr(), out_high.data_ptr(), sp), "k12")
    # bbox: every coordinate is the selected point's coordinate, and no point lies outside its box
    seg = torch.repeat_interleave(torch.arange(B, device=dev), npts.long())
    lo = torch.full((B, 2), float("inf"), dtype=torch.float64, device=dev).scatter_reduce(0, seg[:, None].expand(-1, 2), xy, "amin")
    hi = torch.full((B, 2), float("-inf"), dtype=torch.float64, device=dev).scatter_reduce(0, seg[:, None].expand(-1, 2), xy, "amax")
    assert torch.equal(out_box[:, :2], lo) and torch.equal(out_box[:, 2:], hi)
    base = pt_off[:-1].long()
    assert torch.equal(xy[base + out_arg[:, 0].long(), 0], out_box[:, 0]) and torch.equal(xy[base + out_arg[:, 3].long(), 1], out_box[:, 3])
    assert bool(((out_arg >= 0) & (out_arg < npts[:, None])).all())
    assert bool(out_high[rows_dup].bool().all())                                      # every planted duplicate is found
    # exact oracle run on 20 000 sampled rows (their boxes gathered on the host)
    rows = torch.sort(torch.randperm(N, generator=torch.Generator().manual_seed(5))[:20000]).values.to(dev)
    cnt = nbox[rows].long()
    bidx = torch.repeat_interleave(box_off[rows].long(), cnt) + (torch.arange(int(cnt.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    sb = out_box[bidx].cpu().numpy()
    soff = np.zeros(len(rows) + 1, np.int32)
    np.cumsum(cnt.cpu().numpy(), out=soff[1:])
    assert np.array_equal(out_high[rows].cpu().numpy(), olib.iou_any_ge(sb, soff, 2, 0.98))
    high_share = float(out_high.float().mean())
    assert 0.002 < high_share < 0.05 and int(out_high.sum()) >= rows_dup.numel()
    del xy, out_arg, lo, hi, seg

    # ---- K6 on one record per box ------------------------------------------------------------------------------
    labels = torch.randint(0, 20, (B,), generator=g, device=dev, dtype=torch.int32)
    cat = torch.where(labels < 10, 0, torch.where(labels < 18, 1, -1)).to(torch.int32).contiguous()
    sizes = [int((cat == c).sum()) for c in (0, 1)]
    perm = torch.from_numpy(np.concatenate([native.mt19937_permutation(42, s) for s in sizes])).to(dev)
    cat_off = torch.tensor([0, sizes[0], sizes[0] + sizes[1]], dtype=torch.int64, device=dev)
    n_train = torch.tensor([int(s * 0.8) for s in sizes], dtype=torch.int64, device=dev)
    n_val = torch.tensor([int(s * 0.1) for s in sizes], dtype=torch.int64, device=dev)
    split = torch.empty(B, dtype=torch.uint8, device=dev)
    pos = torch.empty(B, dtype=torch.int64, device=dev)
    ck(L.dyd_split_ids_dev(cat.data_ptr(), B, perm.data_ptr(), cat_off.data_ptr(), n_train.data_ptr(), n_val.data_ptr(), 2,
                           split.data_ptr(), pos.data_ptr(), sp), "k6")
    assert bool((split[cat < 0] == 255).all()) and bool((pos[cat < 0] == -1).all())
    for c in (0, 1):
        m = cat == c
        rank = torch.arange(sizes[c], device=dev)
        assert torch.equal(perm[int(cat_off[c]):int(cat_off[c + 1])][pos[m]], rank)   # perm[position] == rank in the category
        tr, va = int(n_train[c]), int(n_val[c])
        want = torch.where(pos[m] < tr, 0, torch.where(pos[m] < tr + va, 1, 2)).to(torch.uint8)
        assert torch.equal(split[m], want)
    # ---- K8 + K6: the permutations made on the device from the seed must place all 165 M records exactly where the host's
    # sequential Fisher-Yates (numpy's own algorithm) does ------------------------------------------------------------------
    split8 = torch.empty(B, dtype=torch.uint8, device=dev)
    pos8 = torch.empty(B, dtype=torch.int64, device=dev)
    h_sizes = np.asarray(sizes, np.int64)
    h_tr, h_va = n_train.cpu().numpy(), n_val.cpu().numpy()
    ck(L.dyd_split_ids_seeded_dev(cat.data_ptr(), B, 42, h_sizes.ctypes.data, h_tr.ctypes.data, h_va.ctypes.data, 2, None,
                                  split8.data_ptr(), pos8.data_ptr(), sp), "k8 + k6")
    assert torch.equal(pos8, pos) and torch.equal(split8, split)
    assert L.dyd_device_status(None) == 0


def test_config4_dense_256_boxes_per_image(native):
    torch, dev = _dev()
    L, ck = native.lib(), native.check
    sp = torch.cuda.current_stream().cuda_stream
    N, PER, CHUNK = 10_000_000, 256, 5_000_000
    g = torch.Generator(device=dev).manual_seed(7)
    row_off = (torch.arange(CHUNK + 1, device=dev, dtype=torch.int64) * PER).to(torch.int32)
    total_high = 0
    for c0 in range(0, N, CHUNK):
        B = CHUNK * PER
        box = torch.empty((B, 4), dtype=torch.float64, device=dev)
        for part in range(0, B, B // 8):                                             # generated in pieces: rand + temporaries
            n = B // 8
            ctr = torch.rand((n, 2), generator=g, device=dev, dtype=torch.float64) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)
            half = torch.rand((n, 2), generator=g, device=dev, dtype=torch.float64) * 40 + 5
            box[part:part + n, :2] = torch.round(ctr - half)
            box[part:part + n, 2:] = torch.round(ctr + half)
            del ctr, half
        # rows r with r % 50 == 0: box 255 := box 3 with its top edge 1 % lower -> IoU 0.99
        planted = torch.arange(0, CHUNK, 50, device=dev)
        src, dst = planted * PER + 3, planted * PER + 255
        box[dst] = box[src]
        box[dst, 1] += torch.floor((box[src, 3] - box[src, 1]) * 0.01 * 100) / 100
        high = torch.empty(CHUNK, dtype=torch.uint8, device=dev)
        ck(L.dyd_iou_any_ge_dev(box.data_ptr(), row_off.data_ptr(), CHUNK, int(box.shape[0]), 2, 0.98, high.data_ptr(), None, sp), "k2")
        assert bool(high[planted].bool().all())
        rows = torch.sort(torch.randperm(CHUNK, generator=torch.Generator().manual_seed(c0 + 1))[:1500]).values.to(dev)
        rows = torch.cat([rows, planted[:100]])
        sb = box.reshape(CHUNK, PER, 4)[rows].reshape(-1, 4).cpu().numpy()
        soff = (np.arange(len(rows) + 1) * PER).astype(np.int32)
        assert np.array_equal(high[rows].cpu().numpy(), olib.iou_any_ge(sb, soff, 2, 0.98))   # exact on the sample
        # thr = 0 makes every row with two boxes HIGH (inter == 0 -> 0.0 >= 0 holds, processor.py:334-335)
        ck(L.dyd_iou_any_ge_dev(box.data_ptr(), row_off.data_ptr(), CHUNK, int(box.shape[0]), 2, 0.0, high.data_ptr(), None, sp), "k2")
        assert bool(high.bool().all())
        ck(L.dyd_iou_any_ge_dev(box.data_ptr(), row_off.data_ptr(), CHUNK, int(box.shape[0]), 257, 0.0, high.data_ptr(), None, sp), "k2")
        assert not bool(high.bool().any())                                            # fewer boxes than min_boxes
        total_high += 1
        del box, high
    assert total_high == N // CHUNK
