#!/usr/bin/env python3
"""One dense table (256 boxes per image), a handful of K2 launches per mode — the command to put under rocprofv3 --pmc when the
dense kernels need explaining.  Modes: flag98 / flag30 (flag at that threshold), thr<value>, max (diagnostic maximum at 0.98).
    python tools/k2_dense_probe.py [--boxes 64000000] [--bpr 256] [--variant 5] [--iters 3]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boxes", type=int, default=64_000_000)
    ap.add_argument("--bpr", type=int, default=256)
    ap.add_argument("--variant", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--modes", default="flag98,flag30,max")
    ap.add_argument("--dup", type=float, default=0.05, help="share of rows whose last box nearly repeats the first")
    ap.add_argument("--tie", type=float, default=0.001, help="share of rows ending in the exact-tie pair")
    args = ap.parse_args()
    import torch
    from deal_yolo_daya_amd import _native, synth
    L = _native.lib()
    dev = torch.device("cuda:0")
    sp = torch.cuda.current_stream().cuda_stream
    ck = _native.check
    n = args.boxes // args.bpr
    d = synth.generate_device(n, 7, dev, boxes_per_row=args.bpr, dup_prob=args.dup, tie_prob=args.tie)
    xy, pt_off, box_off = d["xy"], d["pt_off"], d["box_off"]
    B, P = int(pt_off.shape[0]) - 1, int(xy.shape[0])
    out_box = torch.empty((B, 4), dtype=torch.float64, device=dev); out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
    out_high = torch.empty(n, dtype=torch.uint8, device=dev); mx = torch.empty(n, dtype=torch.float64, device=dev)
    ck(L.dyd_bbox_minmax_dev(xy.data_ptr(), pt_off.data_ptr(), B, P, out_box.data_ptr(), out_arg.data_ptr(), sp), "k1")
    ck(L.dyd_set_option(b"k2_variant", args.variant), "opt")
    res = {}
    dbg = None
    if L.dyd_set_option(b"k2s_debug_ptr", 0) == 0:   # an experiment build of the library (-DK2S_DEBUG, DYD_LIB_PATH): sweep counters
        dbg = torch.zeros(16, dtype=torch.int64, device=dev)
        ck(L.dyd_set_option(b"k2s_debug_ptr", dbg.data_ptr()), "dbg")
    for mode in args.modes.split(","):
        thr = 0.3 if mode == "flag30" else (float(mode[3:]) if mode.startswith("thr") else 0.98)
        ts = []
        for _ in range(args.iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ck(L.dyd_iou_any_ge_dev(out_box.data_ptr(), box_off.data_ptr(), n, B, 2, thr, out_high.data_ptr(),
                                    mx.data_ptr() if mode == "max" else None, sp), "k2")
            b.record(); b.synchronize()
            ts.append(round(a.elapsed_time(b), 4))
        res[mode] = ts
        if dbg is not None:
            c = dbg.cpu().tolist(); dbg.zero_()
            k = max(1, c[0])
            res[mode + "_counters"] = {"rows_swept": c[0] // args.iters, "rows_not_finite": c[1] // args.iters, "iterations_per_row": round(c[2] / k, 2),
                                       "candidates_per_row": round(c[3] / k, 2), "drains_per_row": round(c[4] / k, 3), "max_iterations": c[5]}
    print(json.dumps({"rows": n, "boxes_per_row": args.bpr, "variant": args.variant, "dup": args.dup, "tie": args.tie, "ms": res}))


if __name__ == "__main__":
    main()
