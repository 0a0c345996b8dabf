#!/usr/bin/env python3
"""Randomised soak of the device kernels against the CPU oracle: fresh shapes and seeds every round, through the host entry
points of the C ABI (K1, K2, K4, K5, K6, K7, K8), the fused device entry and the product's replace -> IoU pass on JSON cells.  Stops at the first mismatch.
    python tools/soak.py --seconds 240 [--seed N]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def same_f64(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint64), b[~np.isnan(b)].view(np.uint64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=int(time.time()) & 0xffffff)
    a = ap.parse_args()
    import torch
    from helpers import random_boxes, random_polygons
    from deal_yolo_daya_amd import _native
    from oracle import lib as olib
    import test_gpu_yolo as ty
    L = _native.lib()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    rounds, counts = 0, {}

    def bump(name):
        counts[name] = counts.get(name, 0) + 1

    while time.time() < t_end:
        rounds += 1
        seed = int(rng.integers(0, 2 ** 31))
        r = np.random.default_rng(seed)
        what = rounds % 10
        ctx = {"round": rounds, "seed": seed, "what": what}
        try:
            if what == 0:      # K1, all three kernels
                n, mp = int(r.integers(1, 60000)), int(r.choice([3, 12, 30, 60, 300, 3000]))
                n = min(n, 4_000_000 // mp + 1)
                xy, off = random_polygons(r, n, mp)
                obox, oarg = olib.bbox_minmax(xy, off)
                for v in (-1, 0, 2):
                    _native.check(L.dyd_set_option(b"k1_variant", v), "opt")
                    box, arg = _native.bbox_minmax(xy, off)
                    assert np.array_equal(arg, oarg) and same_f64(box, obox), ("k1", v)
                _native.check(L.dyd_set_option(b"k1_variant", -1), "opt")
                bump("k1")
            elif what == 1:    # K2, small and big rows, with the maximum
                fixed = int(r.choice([0, 0, 0, 200, 700, 3000]))
                n_rows = int(r.integers(1, 3000)) if not fixed else int(r.integers(1, max(2, 6000 // fixed)))
                box, off = random_boxes(r, n_rows, int(r.integers(1, 90)), True, fixed or None)
                thr, mb = float(r.choice([0.98, 0.5, 0.0, 1.0, 0.9])), int(r.choice([2, 2, 3, 1]))
                want, wmx = olib.iou_any_ge(box, off, mb, thr, want_max=True)
                for v in (-1, 3, 0, 1, 2, 4, 5):
                    _native.check(L.dyd_set_option(b"k2_variant", v), "opt")
                    got, gmx = _native.iou_any_ge(box, off, mb, thr, want_max=True)
                    assert np.array_equal(got, want) and np.array_equal(gmx.view(np.uint64), wmx.view(np.uint64)), ("k2", v)
                _native.check(L.dyd_set_option(b"k2_variant", -1), "opt")
                bump("k2")
            elif what == 2:    # fused device entry, every variant
                n_rows = int(r.integers(1, 4000))
                mb_ = int(r.choice([4, 32, 70, 150]))
                bx, boff = random_boxes(r, n_rows, mb_, False)
                B = int(boff[-1])
                xy, poff = random_polygons(r, B, int(r.choice([5, 12, 40, 120])))
                obox, oarg, want = olib.bbox_iou_chain(xy, poff, boff, 2, 0.5)
                dev = torch.device("cuda:0")
                t_xy = torch.from_numpy(np.ascontiguousarray(xy)).to(dev); t_po = torch.from_numpy(poff).to(dev); t_bo = torch.from_numpy(boff).to(dev)
                t_box = torch.empty((B, 4), dtype=torch.float64, device=dev); t_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
                t_high = torch.empty(n_rows, dtype=torch.uint8, device=dev)
                for v in (-1, 1, 4, 6, 9, 10):
                    _native.check(L.dyd_set_option(b"fused_variant", v), "opt")
                    t_box.fill_(-7.0); t_arg.fill_(-7); t_high.fill_(9)
                    _native.check(L.dyd_bbox_iou_fused_dev(t_xy.data_ptr(), t_po.data_ptr(), t_bo.data_ptr(), n_rows, B, int(xy.shape[0]), 2, 0.5,
                                                           t_box.data_ptr(), t_arg.data_ptr(), t_high.data_ptr(), torch.cuda.current_stream().cuda_stream), "fused")
                    torch.cuda.synchronize()
                    assert np.array_equal(t_arg.cpu().numpy(), oarg) and same_f64(t_box.cpu().numpy(), obox), ("fused k1", v)
                    assert np.array_equal(t_high.cpu().numpy(), want), ("fused k2", v)
                _native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
                bump("fused")
            elif what == 3:    # K4 / K5 on keys with many duplicates and forced slot collisions
                n = int(r.integers(1, 200000))
                distinct = int(r.choice([1, 3, 50, n // 3 + 1, n]))
                keys = r.integers(0, 2 ** 63, (distinct, 2), dtype=np.int64).astype(np.uint64)
                if r.random() < 0.3:
                    keys[:, 0] &= np.uint64(0xff)                # same home slots
                h = keys[r.integers(0, distinct, n)]
                for keep in (0, 1, 2):
                    assert np.array_equal(_native.dedup(h, {0: "first", 1: "last", 2: False}[keep]), olib.dedup(h, keep)), ("k4", keep)
                ref = np.concatenate([keys[: max(1, distinct // 2)], r.integers(0, 2 ** 63, (int(r.integers(1, 5000)), 2), dtype=np.int64).astype(np.uint64)])
                assert np.array_equal(_native.isin(h, ref), olib.isin(h, ref)), "k5"
                bump("k4k5")
            elif what == 4:    # K6
                n, n_cat = int(r.integers(1, 300000)), int(r.choice([1, 2, 7, 300, 2000]))
                cat = r.integers(-1, n_cat, n).astype(np.int32)
                sizes = np.bincount(cat[cat >= 0], minlength=n_cat)
                perm = np.concatenate([r.permutation(int(s)) for s in sizes]).astype(np.int64) if n_cat else np.zeros(0, np.int64)
                cat_off = np.zeros(n_cat + 1, np.int64); np.cumsum(sizes, out=cat_off[1:])
                ntr, nva = (sizes * 8 // 10).astype(np.int64), (sizes // 10).astype(np.int64)
                got = _native.split_ids(cat, perm, cat_off, ntr, nva)
                want = olib.split_ids(cat, perm, cat_off, ntr, nva)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), "k6"
                bump("k6")
            elif what == 7:    # K8 + K6 from a seed: permutations of random sizes against numpy and the oracle's sequential loop
                n = int(r.choice([1, 2, 3, 700, 40000, 33000, 250000, 1200000]))
                n = int(r.integers(max(1, n // 2), n + 1))
                seed8 = int(r.integers(0, 2 ** 32))
                perm, inv = _native.mt19937_permutation_device(seed8, n, want_inverse=True)
                assert np.array_equal(perm, np.random.RandomState(seed8).permutation(n)) and np.array_equal(inv[perm], np.arange(n)), ("k8", n)
                n_cat = int(r.choice([1, 2, 5]))
                cat = r.integers(-1, n_cat, int(r.integers(1, 400000))).astype(np.int32)
                sizes = np.bincount(cat[cat >= 0], minlength=n_cat).astype(np.int64)
                perm = np.concatenate([olib.mt19937_permutation(seed8, int(sz)) for sz in sizes])
                cat_off = np.zeros(n_cat + 1, np.int64); np.cumsum(sizes, out=cat_off[1:])
                ntr, nva = (sizes * 8 // 10).astype(np.int64), (sizes // 10).astype(np.int64)
                got = _native.split_ids_seeded(cat, seed8, sizes, ntr, nva)
                want = olib.split_ids(cat, perm, cat_off, ntr, nva)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), "k8+k6"
                bump("k8")
            elif what == 8:    # the product's replace -> IoU pass on JSON cells: native pipeline == stepwise route == chain oracle
                import pandas as pd
                from deal_yolo_daya_amd import synth
                from deal_yolo_daya_amd.core import processor as P
                t = synth.generate(int(r.integers(1, 6000)), seed=seed, max_boxes=int(r.choice([2, 8, 32, 90])))
                if r.random() < 0.5:          # plant empty polygons: the chain's prefix rule
                    npts = np.diff(t.pt_off)
                    kill = r.random(len(npts)) < 0.05
                    keep_pt = np.repeat(~kill, npts)
                    t.xy = t.xy[keep_pt]
                    npts = np.where(kill, 0, npts)
                    t.pt_off = np.concatenate([[0], np.cumsum(npts)]).astype(np.int32)
                df = pd.DataFrame({"source": synth.urls(t), P.ANNOTATION_COL: synth.json_cells(t)})
                mb, thr = int(r.choice([2, 3])), float(r.choice([0.98, 0.5]))
                a1 = P.replace_and_filter_frame(df, mb, thr)
                os.environ["DYD_NATIVE_PIPELINE"] = "0"
                try:
                    a2 = P.replace_and_filter_frame(df, mb, thr)
                finally:
                    del os.environ["DYD_NATIVE_PIPELINE"]
                _, oarg, ohigh = olib.bbox_iou_chain(t.xy, t.pt_off, t.box_off, mb, thr)
                assert a1[0][P.BBOX_COL].tolist() == a2[0][P.BBOX_COL].tolist(), "pipeline vs stepwise text"
                assert a1[2].index.tolist() == a2[2].index.tolist() == np.flatnonzero(ohigh).tolist(), "pipeline / stepwise / oracle HIGH rows"
                bump("replace_iou")
            elif what == 9:    # rows of 33..300 boxes (sort and sweep, k2_sweep.h): K2 variants with the maximum, fused variants with the chain rule
                n_rows = int(r.integers(1, 120))
                sizes = r.integers(33, 301, size=n_rows)
                sizes[r.random(n_rows) < 0.15] = r.integers(0, 33, size=int((r.random(n_rows) < 0.15).sum() or 1))[0]
                boff = np.zeros(n_rows + 1, np.int32); np.cumsum(sizes, out=boff[1:])
                B = int(boff[-1])
                style = int(r.integers(0, 5))
                ctr = r.random((B, 2)) * [1920, 1080]
                if style == 1:
                    ctr[:, 0] = np.round(ctr[:, 0] / 120.0) * 120.0          # columns: many boxes share x1
                elif style == 2:
                    ctr = ctr * 0.02 + 500.0                                 # one crowded spot
                elif style == 3:
                    ctr = ctr * 4096.0 + 16777216.0                          # beyond f32 resolution
                wh = r.random((B, 2)) * (1.0 if style == 2 else 90.0) + 1.0
                npts = r.integers(1, 6, size=B)
                npts[r.random(B) < 0.01] = 0                                 # polygons without a valid point
                poff = np.zeros(B + 1, np.int32); np.cumsum(npts, out=poff[1:])
                xy = np.repeat(ctr, npts, axis=0) + (r.random((int(poff[-1]), 2)) - 0.5) * np.repeat(wh, npts, axis=0)
                if style != 3:
                    xy = np.round(xy, int(r.integers(0, 3)))
                for rr in range(0, n_rows, 2):                               # a copy of a random box at the end of every other row
                    s0, e0 = int(boff[rr]), int(boff[rr + 1])
                    if e0 - s0 >= 2:
                        src, dst = int(r.integers(s0, e0 - 1)), e0 - 1
                        k = min(int(npts[src]), int(npts[dst]))
                        if k:
                            xy[poff[dst]:poff[dst] + k] = xy[poff[src]:poff[src] + k]
                            xy[poff[dst] + k:poff[dst + 1]] = xy[poff[src]]
                            xy[poff[src] + k:poff[src + 1]] = xy[poff[src]]
                if r.random() < 0.3 and len(xy):
                    xy[r.integers(0, len(xy), 5), r.integers(0, 2, 5)] = r.choice([np.nan, np.inf, -np.inf], 5)
                thr, mb = float(r.choice([0.98, 0.9, 0.5, 0.3, 1.0, 1e-9, 0.0])), int(r.choice([2, 2, 3, 40]))
                obox, oarg, want = olib.bbox_iou_chain(xy, poff, boff, mb, thr)
                for v in (-1, 4, 6, 9, 10):
                    _native.check(L.dyd_set_option(b"fused_variant", v), "opt")
                    arg, high, box = _native.bbox_iou_fused(xy, poff, boff, mb, thr, want_box=True)
                    assert np.array_equal(arg, oarg) and same_f64(box, obox), ("dense fused k1", v, style)
                    assert np.array_equal(high, want), ("dense fused k2", v, style, thr, mb, np.flatnonzero(high != want)[:4].tolist())
                _native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
                kbox = np.where(np.isnan(obox), 0.0, obox)                     # K2 alone on the boxes (empty polygons as zero boxes)
                want2, wmx = olib.iou_any_ge(kbox, boff, mb, thr, want_max=True)
                for v in (-1, 3, 5, 2, 4):
                    _native.check(L.dyd_set_option(b"k2_variant", v), "opt")
                    got, gmx = _native.iou_any_ge(kbox, boff, mb, thr, want_max=True)
                    assert np.array_equal(got, want2) and np.array_equal(gmx.view(np.uint64), wmx.view(np.uint64)), ("dense k2", v, style, thr, mb)
                _native.check(L.dyd_set_option(b"k2_variant", -1), "opt")
                bump("dense_rows")
            else:              # K7, every kernel
                n_rows, mbx = int(r.integers(1, 20000)), int(r.choice([1, 1, 2, 5, 40, 700]))
                n_rows = min(n_rows, 400000 // mbx + 1)
                case = ty._random_case(r, n_rows, mbx, bool(r.integers(0, 2)))
                for v in (-1, 22, 2, 30):
                    _native.check(L.dyd_set_option(b"k7_variant", v), "opt")
                    ty._check_against_oracle(_native, *case)
                _native.check(L.dyd_set_option(b"k7_variant", -1), "opt")
                bump("k7")
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"FAILED": ctx, "error": repr(e)[:500]}), flush=True)
            raise
        if rounds % 50 == 0:
            print(json.dumps({"rounds": rounds, "counts": counts}), flush=True)
    print(json.dumps({"soak": "ok", "seed": a.seed, "rounds": rounds, "counts": counts}), flush=True)


if __name__ == "__main__":
    main()
