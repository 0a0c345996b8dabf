#!/usr/bin/env python3
"""What does this box do for the fused kernel's stream?  k0_membench modes 10-12 move 72 % read / 28 % write and nothing else
(five 16-byte loads in flight per lane, two 16-byte stores), at the fused launch's own footprint (20.5 GB read, 8.2 GB written for
the 10 M-row table); beside them the pure read / write / copy streams and the fused kernel itself on a 10 M-row table, same box,
same process, HIP-event timed, alternating.  One JSON line per measurement."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--read-gb", type=float, default=20.5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows of the fused kernel's table (0 = skip the kernel)")
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    import torch
    from deal_yolo_daya_amd import _native, synth

    dev = torch.device("cuda", 0)
    L = _native.lib()
    ck = _native.check
    sp = torch.cuda.current_stream().cuda_stream
    n_read = int(args.read_gb * 1e9) // 80 * 80
    src = torch.empty(n_read, dtype=torch.uint8, device=dev)
    src.random_(0, 255)
    dst = torch.empty(n_read * 2 // 5, dtype=torch.uint8, device=dev)

    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts)), float(np.min(ts))

    fused = None
    if args.rows:
        xy_p, npts_p, nbox_p = [], [], []
        for ci, start in enumerate(range(0, args.rows, 2_000_000)):
            d = synth.generate_device(min(2_000_000, args.rows - start), synth.SEED + ci, dev)
            xy_p.append(d["xy"]); npts_p.append(torch.diff(d["pt_off"])); nbox_p.append(torch.diff(d["box_off"]))
            del d
        xy = torch.cat(xy_p); npts = torch.cat(npts_p); nbox = torch.cat(nbox_p)
        del xy_p, npts_p, nbox_p
        P, B, N = int(xy.shape[0]), int(npts.shape[0]), int(nbox.shape[0])
        pt_off = torch.zeros(B + 1, dtype=torch.int32, device=dev); pt_off[1:] = torch.cumsum(npts, 0, dtype=torch.int64).to(torch.int32)
        box_off = torch.zeros(N + 1, dtype=torch.int32, device=dev); box_off[1:] = torch.cumsum(nbox, 0, dtype=torch.int64).to(torch.int32)
        del npts, nbox
        out_box = torch.empty((B, 4), dtype=torch.float64, device=dev)
        out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
        out_high = torch.empty(N, dtype=torch.uint8, device=dev)
        alg = 16 * P + 4 * (B + 1) + 48 * B + 4 * (N + 1) + N
        fused = lambda: ck(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, P, 2, 0.98, out_box.data_ptr(),  # noqa: E731
                                                    out_arg.data_ptr(), out_high.data_ptr(), sp), "fused")
        for _ in range(40):                       # clock ramp
            fused()
        torch.cuda.synchronize()

    def report(name, nbytes, med, mn, **kw):
        print(json.dumps({"what": name, "GB": round(nbytes / 1e9, 3), "ms_median": round(med, 4), "ms_min": round(mn, 4),
                          "TBs_median": round(nbytes / med / 1e9, 3), "TBs_best": round(nbytes / mn / 1e9, 3), **kw}), flush=True)

    for rnd in range(2):                           # twice, alternating, so that a drifting clock shows
        if fused:
            med, mn = timeit(fused)
            report("k12_wave_kernel (fused K1+K2), 10 M rows", alg, med, mn, read_share=round((16 * P + 4 * (B + 1) + 4 * (N + 1)) / alg, 3), round=rnd)
        for mode, nm in ((10, "mix 5:2 plain loads, plain stores"), (11, "mix 5:2 plain loads, nt stores"), (12, "mix 5:2 nt loads, nt stores")):
            for blocks in (2048, 4096):
                med, mn = timeit(lambda: ck(L.dyd_membench_dev(mode, src.data_ptr(), dst.data_ptr(), n_read, blocks, sp), "mb"))
                report(nm, n_read + n_read * 2 // 5, med, mn, blocks=blocks, round=rnd)
        for mode, nm, nb in ((1, "read only", n_read), (4, "read only nt", n_read), (2, "write only", n_read * 2 // 5),
                             (5, "write only nt", n_read * 2 // 5)):
            med, mn = timeit(lambda: ck(L.dyd_membench_dev(mode, src.data_ptr(), dst.data_ptr() if mode in (2, 5) else src.data_ptr(), nb, 2048, sp), "mb"))
            report(nm, nb, med, mn, round=rnd)
    print(json.dumps({"device": _native.device_name()}))


if __name__ == "__main__":
    main()
