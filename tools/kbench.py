#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on one MI355X (device-resident inputs, HIP-event timing,
interleaved A/B rounds in one process).  Prints one JSON line per kernel/variant.

    python tools/kbench.py [--rows 1000000] [--iters 20] [--only k1,k2,...]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="k1,k2,k3,k4,k5,k6")
    ap.add_argument("--bpr", type=int, default=0, help="fixed boxes per row (0 = U{1..32})")
    args = ap.parse_args()
    only = set(args.only.split(","))

    import torch
    from deal_yolo_daya_amd import _native, synth

    dev = torch.device("cuda", 0)
    L = _native.lib()
    sp = torch.cuda.current_stream().cuda_stream
    ck = _native.check

    def timeit(fn, iters=args.iters, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts)), float(np.min(ts))

    def timeit_b2b(fn, n=100, warm=10):
        """ms per launch when the launches are queued back to back (what bench.py times): one event pair around n launches"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n

    def report(name, nbytes, med, mn, **kw):
        print(json.dumps({"kernel": name, "ms_median": round(med, 4), "ms_min": round(mn, 4),
                          "alg_GB": round(nbytes / 1e9, 4), "GBs_median": round(nbytes / med / 1e6, 1),
                          "frac_of_8TBs": round(nbytes / med / 1e6 / 8000, 4), **kw}), flush=True)

    rows = args.rows
    parts = []
    for ci, s in enumerate(range(0, rows, 1_000_000)):
        parts.append(synth.generate(min(1_000_000, rows - s), seed=synth.SEED + ci,
                                    boxes_per_row=args.bpr or None))
    xy = torch.cat([torch.from_numpy(t.xy) for t in parts]).to(dev)
    npts = torch.cat([torch.from_numpy(np.diff(t.pt_off)) for t in parts]).to(dev)
    nbox = torch.cat([torch.from_numpy(np.diff(t.box_off)) for t in parts]).to(dev)
    url_id = np.concatenate([t.url_id for t in parts])
    labels = torch.cat([torch.from_numpy(t.label) for t in parts]).to(dev)
    del parts
    P, B, N = xy.shape[0], npts.shape[0], nbox.shape[0]
    pt_off = torch.zeros(B + 1, dtype=torch.int32, device=dev); pt_off[1:] = torch.cumsum(npts, 0).to(torch.int32)
    box_off = torch.zeros(N + 1, dtype=torch.int32, device=dev); box_off[1:] = torch.cumsum(nbox, 0).to(torch.int32)
    out_box = torch.empty((B, 4), dtype=torch.float64, device=dev)
    out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
    out_high = torch.empty(N, dtype=torch.uint8, device=dev)
    print(json.dumps({"rows": N, "boxes": B, "points": P, "device": _native.device_name()}), flush=True)

    if "k1" in only:
        k1_bytes = 16 * P + 4 * (B + 1) + 48 * B
        res = {}
        for rnd in range(2):                      # interleaved rounds (guide rule 24)
            for variant in (0, 1):
                ck(L.dyd_set_option(b"k1_variant", variant), "opt")
                med, mn = timeit(lambda: ck(L.dyd_bbox_minmax_dev(xy.data_ptr(), pt_off.data_ptr(), B, P,
                                                                   out_box.data_ptr(), out_arg.data_ptr(), sp), "k1"))
                res.setdefault(variant, []).append((med, mn))
        ck(L.dyd_set_option(b"k1_variant", -1), "opt")
        for variant, name in ((0, "k1_bbox_lds"), (1, "k1_bbox_direct")):
            med = float(np.median([r[0] for r in res[variant]])); mn = min(r[1] for r in res[variant])
            report(name, k1_bytes, med, mn, rows_per_s=round(N / med * 1e3))
        # streaming ceilings of this box on a comparable byte count (each array 1.4 GB >> 256 MiB L3)
        nb_ = k1_bytes // 2 // 16 * 16
        src = torch.empty(nb_ // 8, dtype=torch.float64, device=dev).normal_(); dst = torch.empty_like(src)
        for blocks in (2048, 8192):
            for mode, nm, tot in ((0, "copy", 2 * nb_), (1, "read", nb_), (2, "write", nb_), (3, "copy_nt", 2 * nb_),
                                  (4, "read_nt", nb_), (5, "write_nt", nb_)):
                med, mn = timeit(lambda: ck(L.dyd_membench_dev(mode, src.data_ptr(), dst.data_ptr(), nb_, blocks, sp), "mb"))
                report(f"membench_{nm}_b{blocks}", tot, med, mn)
        med, mn = timeit(lambda: dst.copy_(src))
        report("torch_d2d_copy", 2 * nb_, med, mn)
        del src, dst
    if "rand" in only:
        # random-access ceilings (one 8-byte word per lane, every word of the table once): what K4/K5/K6 are quoted against
        for mib in (32, 64, 128, 256, 512, 1024, 2048):          # around the 256 MiB Infinity Cache
            table = torch.zeros(mib << 17, dtype=torch.int64, device=dev)
            for mode, nm, word in ((6, "scatter", 8), (9, "scatter4", 4), (7, "gather", 8), (8, "atomic_min", 8)):
                words = (mib << 20) // word
                med, mn = timeit(lambda: ck(L.dyd_membench_dev(mode, table.data_ptr(), table.data_ptr(), mib << 20, 8192, sp), "mb"))
                print(json.dumps({"kernel": f"membench_random_{nm}_{mib}MiB", "ms_median": round(med, 4), "ms_min": round(mn, 4),
                                  "words": words, "G_words_per_s": round(words / med / 1e6, 2), "useful_GBs": round(word * words / med / 1e6, 1)}), flush=True)
            del table

    if "k2" in only:
        ck(L.dyd_bbox_minmax_dev(xy.data_ptr(), pt_off.data_ptr(), B, P, out_box.data_ptr(), out_arg.data_ptr(), sp), "k1")
        k2_bytes = 32 * B + 4 * (N + 1) + N
        nb64 = nbox.to(torch.int64)
        pairs = int((nb64 * (nb64 - 1) // 2).sum().item())
        for variant, nm in ((0, "k2_iou<16,256>"), (1, "k2_iou<8,128>"), (2, "k2f_iou<16,256> (f32 filter)"),
                            (3, "k2f_iou<8,128> (f32 filter)"), (4, "k2_wave64 (the fused kernel's pair stage)"), (-1, "k2 auto")):
            ck(L.dyd_set_option(b"k2_variant", variant), "opt")
            med, mn = timeit(lambda: ck(L.dyd_iou_any_ge_dev(out_box.data_ptr(), box_off.data_ptr(), N, B, 2, 0.98,
                                                              out_high.data_ptr(), None, sp), "k2"))
            report(nm, k2_bytes, med, mn, rows_per_s=round(N / med * 1e3), pairs=pairs,
                   gpairs_per_s=round(pairs / med / 1e6, 2), high=int(out_high.sum().item()))
        ck(L.dyd_set_option(b"k2_variant", 3), "opt")
        mx = torch.empty(N, dtype=torch.float64, device=dev)
        med, mn = timeit(lambda: ck(L.dyd_iou_any_ge_dev(out_box.data_ptr(), box_off.data_ptr(), N, B, 2, 0.98,
                                                          out_high.data_ptr(), mx.data_ptr(), sp), "k2max"))
        report("k2_iou_want_max", k2_bytes + 8 * N, med, mn, gpairs_per_s=round(pairs / med / 1e6, 2))

    if "k12" in only:
        k12_bytes = 16 * P + 4 * (B + 1) + 48 * B + 4 * (N + 1) + N
        res = {}
        for rnd in range(2):
            for variant in (1, 4, 6, 9, 10):
                ck(L.dyd_set_option(b"fused_variant", variant), "opt")
                med, mn = timeit(lambda: ck(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(),
                                                                      N, B, P, 2, 0.98, out_box.data_ptr(), out_arg.data_ptr(),
                                                                      out_high.data_ptr(), sp), "k12"))
                res.setdefault(variant, []).append((med, mn))
        ck(L.dyd_set_option(b"fused_variant", -1), "opt")
        for variant, name in ((4, "k12_wave_kernel"), (10, "k12_wave_dense_kernel"), (6, "k12_fused<1024,8,128,filter>"),
                              (9, "k12_fused<1024,8,256,filter>"), (1, "k1_then_k2_two_launches")):
            med = float(np.median([r[0] for r in res[variant]])); mn = min(r[1] for r in res[variant])
            report(name, k12_bytes, med, mn, rows_per_s=round(N / med * 1e3), high=int(out_high.sum().item()))

    if only & {"k3", "k4", "k5"}:
        urls = [f"http://img.example/{k}.jpg".encode() for k in url_id.tolist()]
        off_np = np.zeros(N + 1, np.int64); np.cumsum([len(u) for u in urls], out=off_np[1:])
        data = torch.from_numpy(np.frombuffer(b"".join(urls), np.uint8).copy()).to(dev)
        off = torch.from_numpy(off_np).to(dev)
        h = torch.empty((N, 2), dtype=torch.int64, device=dev)
        k3_bytes = int(off_np[-1]) + 8 * (N + 1) + 16 * N
        med, mn = timeit(lambda: ck(L.dyd_hash128_dev(data.data_ptr(), off.data_ptr(), N, h.data_ptr(), sp), "k3"))
        if "k3" in only:
            report("k3_hash128", k3_bytes, med, mn, rows_per_s=round(N / med * 1e3))
        keep = torch.empty(N, dtype=torch.uint8, device=dev)
        U = len(np.unique(url_id))
        if "k4" in only:
            for mode, nm in ((0, "first"), (1, "last"), (2, "none")):
                med, mn = timeit(lambda: ck(L.dyd_dedup_dev(h.data_ptr(), N, mode, keep.data_ptr(), sp), "k4"))
                report(f"k4_dedup_{nm}", 16 * N + N + 48 * U, med, mn, rows_per_s=round(N / med * 1e3),
                       kept=int(keep.sum().item()), distinct=U)
        if "k5" in only:
            R = max(1, N // 10)
            ref = h[torch.randperm(N, device=dev)[:R]].contiguous()
            med, mn = timeit(lambda: ck(L.dyd_isin_dev(h.data_ptr(), N, ref.data_ptr(), R, keep.data_ptr(), sp), "k5"))
            report("k5_isin", 16 * N + N + 16 * R, med, mn, rows_per_s=round(N / med * 1e3), hits=int(keep.sum().item()))

    if "k6" in only:
        E = B
        cat = torch.where(labels < 10, 0, torch.where(labels < 18, 1, -1)).to(torch.int32).contiguous()
        sizes = [int((cat == c).sum().item()) for c in (0, 1)]
        perm = torch.from_numpy(np.concatenate([_native.mt19937_permutation(42, s) for s in sizes])).to(dev)
        cat_off = torch.tensor([0, sizes[0], sizes[0] + sizes[1]], dtype=torch.int64, device=dev)
        n_train = torch.tensor([int(s * 0.8) for s in sizes], dtype=torch.int64, device=dev)
        n_val = torch.tensor([int(s * 0.1) for s in sizes], dtype=torch.int64, device=dev)
        split = torch.empty(E, dtype=torch.uint8, device=dev); pos = torch.empty(E, dtype=torch.int64, device=dev)
        med, mn = timeit(lambda: ck(L.dyd_split_ids_dev(cat.data_ptr(), E, perm.data_ptr(), cat_off.data_ptr(),
                                                         n_train.data_ptr(), n_val.data_ptr(), 2, split.data_ptr(),
                                                         pos.data_ptr(), sp), "k6"))
        report("k6_split_ids", 21 * E, med, mn, expanded_rows=E, rows_per_s=round(E / med * 1e3))

    if "k3len" in only:
        # K3 against the cell length (signed CDN URLs run to hundreds of bytes): ~1 GB of text per table
        g = torch.Generator(device=dev).manual_seed(8)
        for L_ in (30, 100, 400, 2000):
            n3 = max(1000, 1_000_000_000 // L_)
            lens = torch.randint(max(1, L_ // 2), L_ * 3 // 2 + 1, (n3,), generator=g, device=dev)
            off3 = torch.zeros(n3 + 1, dtype=torch.int64, device=dev)
            off3[1:] = torch.cumsum(lens, 0)
            tot3 = int(off3[-1].item())
            data3 = torch.randint(32, 127, (tot3,), generator=g, device=dev, dtype=torch.uint8)
            h3 = torch.empty((n3, 2), dtype=torch.int64, device=dev)
            med, mn = timeit(lambda: ck(L.dyd_hash128_dev(data3.data_ptr(), off3.data_ptr(), n3, h3.data_ptr(), sp), "k3"), iters=8, warm=2)
            report(f"k3_hash128_len{L_}", tot3 + 8 * (n3 + 1) + 16 * n3, med, mn, rows=n3, text_GB=round(tot3 / 1e9, 3))
            del lens, off3, data3, h3

    if "k6cats" in only:
        # K6 against the number of categories (the columns of the rules sheet): 16.5 M expanded rows, uniform labels
        g = torch.Generator(device=dev).manual_seed(4)
        E6 = 16_500_000
        for ncat in (2, 16, 128, 1000, 5000):
            cat6 = torch.randint(-1, ncat, (E6,), generator=g, device=dev, dtype=torch.int32)
            sizes6 = torch.bincount(cat6[cat6 >= 0].to(torch.int64), minlength=ncat)
            cat_off6 = torch.zeros(ncat + 1, dtype=torch.int64, device=dev)
            cat_off6[1:] = torch.cumsum(sizes6, 0)
            perm6 = torch.cat([torch.randperm(int(sz), generator=g, device=dev) for sz in sizes6.tolist()]).contiguous() if ncat <= 1000 else \
                torch.arange(int(cat_off6[-1].item()), device=dev) - torch.repeat_interleave(cat_off6[:-1], sizes6)
            ntr6 = (sizes6 * 8 // 10).contiguous(); nva6 = (sizes6 // 10).contiguous()
            sp6 = torch.empty(E6, dtype=torch.uint8, device=dev); pos6 = torch.empty(E6, dtype=torch.int64, device=dev)
            med, mn = timeit(lambda: ck(L.dyd_split_ids_dev(cat6.data_ptr(), E6, perm6.data_ptr(), cat_off6.data_ptr(), ntr6.data_ptr(),
                                                             nva6.data_ptr(), ncat, sp6.data_ptr(), pos6.data_ptr(), sp), "k6"), iters=8, warm=2)
            report(f"k6_split_ids_{ncat}_categories", 21 * E6, med, mn, expanded_rows=E6)
            del cat6, perm6, sp6, pos6

    if "k5big" in only:
        # the 10 M-row pipeline's K4 / K5 on random 128-bit keys: 10 M rows (60 % distinct), 1 M reference keys, 10 % hits
        Nk = 10_000_000
        g = torch.Generator(device=dev).manual_seed(5)
        base = torch.randint(-2**62, 2**62, (6_000_000, 2), generator=g, device=dev, dtype=torch.int64)
        hk = base[torch.randint(0, base.shape[0], (Nk,), generator=g, device=dev)].contiguous()
        refk = torch.cat([base[:100_000], torch.randint(-2**62, 2**62, (900_000, 2), generator=g, device=dev, dtype=torch.int64)]).contiguous()
        keepk = torch.empty(Nk, dtype=torch.uint8, device=dev)
        med, mn = timeit(lambda: ck(L.dyd_dedup_dev(hk.data_ptr(), Nk, 0, keepk.data_ptr(), sp), "k4"))
        report("k4_dedup_first_10M", 16 * Nk + Nk + 48 * 6_000_000, med, mn, rows_per_s=round(Nk / med * 1e3), kept=int(keepk.sum().item()))
        med, mn = timeit(lambda: ck(L.dyd_isin_dev(hk.data_ptr(), Nk, refk.data_ptr(), refk.shape[0], keepk.data_ptr(), sp), "k5"))
        report("k5_isin_10M_vs_1M", 16 * Nk + Nk + 16 * refk.shape[0], med, mn, rows_per_s=round(Nk / med * 1e3), hits=int(keepk.sum().item()))
        # the other extreme: one key (a constant column, or all NaN), and 1000 keys — every row contends for few slots
        for distinct in (1, 1000):
            few = base[torch.randint(0, distinct, (Nk,), generator=g, device=dev)].contiguous()
            for mode, nm in ((0, "first"), (1, "last"), (2, "none")):
                med, mn = timeit(lambda: ck(L.dyd_dedup_dev(few.data_ptr(), Nk, mode, keepk.data_ptr(), sp), "k4"), iters=5, warm=1)
                report(f"k4_dedup_{nm}_10M_rows_{distinct}_keys", 16 * Nk + Nk, med, mn, rows_per_s=round(Nk / med * 1e3), kept=int(keepk.sum().item()))
            del few
        del base, hk, refk, keepk

    if "k6big" in only:
        # the 10 M-row pipeline's K6 (165 M expanded rows): the inverse-permutation table no longer fits the Infinity Cache
        E = int(os.environ.get("K6_ROWS", 165_000_000))
        g = torch.Generator(device=dev).manual_seed(3)
        lab = torch.randint(0, 20, (E,), generator=g, device=dev, dtype=torch.int32)
        cat = torch.where(lab < 10, 0, torch.where(lab < 18, 1, -1)).to(torch.int32).contiguous()
        del lab
        sizes = [int((cat == c).sum().item()) for c in (0, 1)]
        perm = torch.cat([torch.randperm(s, generator=g, device=dev) for s in sizes]).contiguous()
        cat_off = torch.tensor([0, sizes[0], sizes[0] + sizes[1]], dtype=torch.int64, device=dev)
        n_train = torch.tensor([int(s * 0.8) for s in sizes], dtype=torch.int64, device=dev)
        n_val = torch.tensor([int(s * 0.1) for s in sizes], dtype=torch.int64, device=dev)
        split = torch.empty(E, dtype=torch.uint8, device=dev); pos = torch.empty(E, dtype=torch.int64, device=dev)
        ref = None
        for mib in [int(v) for v in os.environ.get("K6V", "0,1,0,1").split(",")]:   # 0 = 64-bit inverse table, 1 = 32-bit (default)
            ck(L.dyd_set_option(b"k6_variant", mib), "opt")
            med, mn = timeit(lambda: ck(L.dyd_split_ids_dev(cat.data_ptr(), E, perm.data_ptr(), cat_off.data_ptr(), n_train.data_ptr(),
                                                             n_val.data_ptr(), 2, split.data_ptr(), pos.data_ptr(), sp), "k6"), iters=10, warm=2)
            chk = (int(pos.sum().item()), int(split.to(torch.int64).sum().item()))
            ref = ref or chk
            report(f"k6_split_ids_165M_{'32' if mib else '64'}bit_inverse", 21 * E, med, mn, expanded_rows=E, same_as_first=(chk == ref))
        ck(L.dyd_set_option(b"k6_variant", 1), "opt")
        del cat, perm, split, pos

    if "k7" in only:
        import ctypes as C
        # the shape of a split sheet: one labelled box per expanded row; boxes = K1's output for the synthetic polygons
        ck(L.dyd_bbox_minmax_dev(xy.data_ptr(), pt_off.data_ptr(), B, P, out_box.data_ptr(), out_arg.data_ptr(), sp), "k1")
        E = B
        one = torch.arange(E + 1, dtype=torch.int32, device=dev)
        w = torch.full((E,), 1920.0, dtype=torch.float64, device=dev); h = torch.full((E,), 1080.0, dtype=torch.float64, device=dev)
        cid = (torch.arange(E, device=dev, dtype=torch.int32) % 20).contiguous()
        toff = torch.empty(E + 1, dtype=torch.int64, device=dev); flag = torch.empty(E, dtype=torch.uint8, device=dev)
        total = C.c_int64()
        ck(L.dyd_yolo_lines_dev(out_box.data_ptr(), one.data_ptr(), None, w.data_ptr(), h.data_ptr(), cid.data_ptr(), E, E,
                                toff.data_ptr(), flag.data_ptr(), None, 0, C.byref(total), sp), "k7 measure")
        T = total.value
        text = torch.empty(T, dtype=torch.uint8, device=dev)
        modes = os.environ.get("K7MODE", "full,measure").split(",")
        for variant in [int(v) for v in os.environ.get("K7V", "2,22,30,-1").split(",")]:
            ck(L.dyd_set_option(b"k7_variant", variant), "opt")
            if "full" in modes:
              med, mn = timeit(lambda: ck(L.dyd_yolo_lines_dev(out_box.data_ptr(), one.data_ptr(), None, w.data_ptr(), h.data_ptr(),
                                                              cid.data_ptr(), E, E, toff.data_ptr(), flag.data_ptr(), text.data_ptr(), T,
                                                              C.byref(total), sp), "k7"))
              report(f"k7_yolo_lines_rpt{variant}", 32 * E + 4 * (E + 1) + 20 * E + 8 * (E + 1) + E + T, med, mn, rows=E, text_bytes=T,
                   rows_per_s=round(E / med * 1e3), no_line_rows=int((flag == 1).sum().item()))
            if "measure" not in modes:
                continue
            med, mn = timeit(lambda: ck(L.dyd_yolo_lines_dev(out_box.data_ptr(), one.data_ptr(), None, w.data_ptr(), h.data_ptr(),
                                                              cid.data_ptr(), E, E, toff.data_ptr(), flag.data_ptr(), None, 0,
                                                              C.byref(total), sp), "k7"))
            report(f"k7_measure_only_rpt{variant}", 32 * E + 4 * (E + 1) + 20 * E + 8 * (E + 1) + E, med, mn, rows=E)
        # rows of many boxes (unsplit sheets: 1..32 lines per row): the generic path of the kernel
        wr = torch.full((N,), 1920.0, dtype=torch.float64, device=dev); hr = torch.full((N,), 1080.0, dtype=torch.float64, device=dev)
        cidr = (torch.arange(N, device=dev, dtype=torch.int32) % 20).contiguous()
        toffr = torch.empty(N + 1, dtype=torch.int64, device=dev); flagr = torch.empty(N, dtype=torch.uint8, device=dev)
        ck(L.dyd_yolo_lines_dev(out_box.data_ptr(), box_off.data_ptr(), None, wr.data_ptr(), hr.data_ptr(), cidr.data_ptr(), N, B,
                                toffr.data_ptr(), flagr.data_ptr(), None, 0, C.byref(total), sp), "k7 measure")
        Tr = total.value
        textr = torch.empty(Tr, dtype=torch.uint8, device=dev)
        for variant in [int(v) for v in os.environ.get("K7VM", "30,-1").split(",")]:   # 22: 330 ms per launch
            ck(L.dyd_set_option(b"k7_variant", variant), "opt")
            med, mn = timeit(lambda: ck(L.dyd_yolo_lines_dev(out_box.data_ptr(), box_off.data_ptr(), None, wr.data_ptr(), hr.data_ptr(),
                                                              cidr.data_ptr(), N, B, toffr.data_ptr(), flagr.data_ptr(), textr.data_ptr(), Tr,
                                                              C.byref(total), sp), "k7"))
            report(f"k7_yolo_lines_multi_box_rows_v{variant}", 32 * B + 4 * (N + 1) + 20 * N + 8 * (N + 1) + N + Tr, med, mn, rows=N, lines=B,
                   text_bytes=Tr, lines_per_s=round(B / med * 1e3))
        ck(L.dyd_set_option(b"k7_variant", -1), "opt")
    if "k7mix" in only:
        # where the box-tiled kernel overtakes the row kernels: rows of one box with a share of two-box rows mixed in
        import ctypes as C
        ck(L.dyd_bbox_minmax_dev(xy.data_ptr(), pt_off.data_ptr(), B, P, out_box.data_ptr(), out_arg.data_ptr(), sp), "k1")
        g = torch.Generator(device=dev).manual_seed(9)
        total = C.c_int64()
        for share in (0.0, 0.05, 0.1, 0.25, 0.5, 1.0):
            nr = int(B / (1 + share))
            counts = (torch.rand(nr, generator=g, device=dev) < share).to(torch.int32) + 1
            ro = torch.zeros(nr + 1, dtype=torch.int32, device=dev)
            ro[1:] = torch.cumsum(counts, 0, dtype=torch.int64).to(torch.int32)
            nbx = int(ro[-1].item())
            w2 = torch.full((nr,), 1920.0, dtype=torch.float64, device=dev); h2 = torch.full((nr,), 1080.0, dtype=torch.float64, device=dev)
            c2 = (torch.arange(nr, device=dev, dtype=torch.int32) % 20).contiguous()
            to2 = torch.empty(nr + 1, dtype=torch.int64, device=dev); fl2 = torch.empty(nr, dtype=torch.uint8, device=dev)
            ck(L.dyd_set_option(b"k7_variant", 22), "opt")
            ck(L.dyd_yolo_lines_dev(out_box.data_ptr(), ro.data_ptr(), None, w2.data_ptr(), h2.data_ptr(), c2.data_ptr(), nr, nbx,
                                    to2.data_ptr(), fl2.data_ptr(), None, 0, C.byref(total), sp), "k7 measure")
            tx = torch.empty(total.value, dtype=torch.uint8, device=dev)
            res = {}
            for variant in (22, 30, 22, 30):
                ck(L.dyd_set_option(b"k7_variant", variant), "opt")
                med, mn = timeit(lambda: ck(L.dyd_yolo_lines_dev(out_box.data_ptr(), ro.data_ptr(), None, w2.data_ptr(), h2.data_ptr(), c2.data_ptr(),
                                                                  nr, nbx, to2.data_ptr(), fl2.data_ptr(), tx.data_ptr(), total.value, C.byref(total), sp), "k7"),
                                 iters=10, warm=2)
                res.setdefault(variant, []).append(med)
            print(json.dumps({"k7_boxes_per_row": round(nbx / nr, 3), "rows": nr, "lines": nbx, "pair_rows_ms": round(min(res[22]), 4),
                              "box_tiles_ms": round(min(res[30]), 4)}), flush=True)
            del ro, w2, h2, c2, to2, fl2, tx, counts
        ck(L.dyd_set_option(b"k7_variant", -1), "opt")

if __name__ == "__main__":
    main()
