#!/usr/bin/env python3
"""A sparse table with a heavy tail: rows of 1..32 boxes plus a share of rows of 200 boxes — what does the tail cost per kernel choice?
(the sparse wave kernel queues such rows for the drain kernel's sweep; variant 10 sweeps them in place; 6 is the workgroup kernel)
    python tools/heavy_tail_probe.py [--rows 1000000] [--share 0.02] [--long 200]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--share", default="0,0.002,0.02,0.05")
    ap.add_argument("--long", type=int, default=200)
    a = ap.parse_args()
    import torch
    from deal_yolo_daya_amd import _native
    L = _native.lib()
    dev = torch.device("cuda:0")
    sp = torch.cuda.current_stream().cuda_stream
    ck = _native.check
    g = torch.Generator(device=dev).manual_seed(3)
    for share in [float(v) for v in a.share.split(",")]:
        nb = torch.randint(1, 33, (a.rows,), generator=g, device=dev)
        nb[torch.rand(a.rows, generator=g, device=dev) < share] = a.long
        box_off = torch.zeros(a.rows + 1, dtype=torch.int64, device=dev); box_off[1:] = torch.cumsum(nb, 0)
        B = int(box_off[-1])
        npts = torch.randint(3, 13, (B,), generator=g, device=dev)
        pt_off = torch.zeros(B + 1, dtype=torch.int64, device=dev); pt_off[1:] = torch.cumsum(npts, 0)
        P = int(pt_off[-1])
        cx = torch.rand(B, generator=g, device=dev, dtype=torch.float64) * 1920
        cy = torch.rand(B, generator=g, device=dev, dtype=torch.float64) * 1080
        bop = torch.repeat_interleave(torch.arange(B, device=dev), npts)
        xy = torch.rand((P, 2), generator=g, device=dev, dtype=torch.float64) * 100 - 50
        xy[:, 0] += cx[bop]; xy[:, 1] += cy[bop]
        xy = torch.round(xy * 100) / 100
        del bop
        pt32, bo32 = pt_off.to(torch.int32), box_off.to(torch.int32)
        out_box = torch.empty((B, 4), dtype=torch.float64, device=dev); out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
        out_high = torch.empty(a.rows, dtype=torch.uint8, device=dev)
        line = {"rows": a.rows, "share_of_long_rows": share, "long_row_boxes": a.long, "boxes": B, "mean_boxes_per_row": round(B / a.rows, 2),
                "alg_GB": round((16 * P + 52 * B + 5 * a.rows) / 1e9, 3)}
        highs = set()
        for variant in (4, 10, 6, -1, 4, 10, 6, -1):
            ck(L.dyd_set_option(b"fused_variant", variant), "opt")
            ts = []
            for it in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ck(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt32.data_ptr(), bo32.data_ptr(), a.rows, B, P, 2, 0.98, out_box.data_ptr(), out_arg.data_ptr(),
                                            out_high.data_ptr(), sp), "k12")
                e1.record(); e1.synchronize()
                if it >= 2:
                    ts.append(e0.elapsed_time(e1))
            key = {4: "sparse_wave_ms", 10: "dense_wave_ms", 6: "workgroup_ms", -1: "auto_ms"}[variant]
            line[key] = min(line.get(key, 1e9), round(float(np.median(ts)), 4))
            highs.add(int(out_high.sum().item()))
        ck(L.dyd_set_option(b"fused_variant", -1), "opt")
        line["same_flags"] = len(highs) == 1
        print(json.dumps(line), flush=True)
        del xy, pt_off, box_off, out_box, out_arg, out_high, pt32, bo32, nb, npts, cx, cy


if __name__ == "__main__":
    main()
