#!/usr/bin/env python3
"""Which fused K1+K2 kernel for which table: fixed boxes per row, ~4 M boxes per table, variants 4 (wave-autonomous) and 6
(workgroup tiles + f32 filter) and the automatic choice, interleaved in one process.
    python tools/fused_sweep.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from deal_yolo_daya_amd import _native, synth
    L = _native.lib()
    dev = torch.device("cuda:0")
    sp = torch.cuda.current_stream().cuda_stream

    def ck(rc, what):
        _native.check(rc, what)

    for bpr in (4, 16, 24, 32, 40, 48, 64, 96, 128, 256):
        n = max(1000, 4_000_000 // bpr)
        t = synth.generate(n, seed=7, boxes_per_row=bpr)
        xy = torch.from_numpy(t.xy).to(dev); pt_off = torch.from_numpy(t.pt_off).to(dev); box_off = torch.from_numpy(t.box_off).to(dev)
        B, N = t.n_boxes, t.n_rows
        out_box = torch.empty((B, 4), dtype=torch.float64, device=dev); out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
        out_high = torch.empty(N, dtype=torch.uint8, device=dev)
        res, highs = {}, {}
        for variant in (4, 6, 10, -1, 4, 6, 10, -1):
            ck(L.dyd_set_option(b"fused_variant", variant), "opt")
            ts = []
            for it in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ck(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, int(xy.shape[0]), 2, 0.98, out_box.data_ptr(),
                                            out_arg.data_ptr(), out_high.data_ptr(), sp), "k12")
                b.record(); b.synchronize()
                if it >= 2:
                    ts.append(a.elapsed_time(b))
            res.setdefault(variant, []).append(float(np.median(ts)))
            highs[variant] = int(out_high.sum().item())
        ck(L.dyd_set_option(b"fused_variant", -1), "opt")
        print(json.dumps({"boxes_per_row": bpr, "rows": N, "boxes": B, "wave_ms": round(min(res[4]), 4), "workgroup_filter_ms": round(min(res[6]), 4), "wave_dense_ms": round(min(res[10]), 4),
                          "auto_ms": round(min(res[-1]), 4), "same_flags": len(set(highs.values())) == 1}), flush=True)
        del xy, pt_off, box_off, out_box, out_arg, out_high


if __name__ == "__main__":
    main()
