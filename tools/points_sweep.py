#!/usr/bin/env python3
"""Fused K1+K2 against the number of points per polygon (the bench tables have 3..12): ~64 M points per table, rows of 1..32
boxes, every box with exactly P points; variants 4 (wave-autonomous), 6 (workgroup tiles), 1 (K1 launch then K2 launch) and
the automatic choice, interleaved in one process.
    python tools/points_sweep.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from deal_yolo_daya_amd import _native
    L = _native.lib()
    dev = torch.device("cuda:0")
    sp = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    for ppb in (4, 8, 16, 32, 64, 128, 256, 1024):
        B = max(2000, 64_000_000 // ppb)
        nb = torch.randint(1, 33, (B // 16,), generator=g, device=dev)
        box_off = torch.zeros(nb.numel() + 1, dtype=torch.int32, device=dev)
        box_off[1:] = torch.cumsum(nb, 0).to(torch.int32)
        N, B = nb.numel(), int(box_off[-1].item())
        P = B * ppb
        centre = torch.rand((B, 1, 2), generator=g, device=dev, dtype=torch.float64) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)
        xy = (centre + torch.rand((B, ppb, 2), generator=g, device=dev, dtype=torch.float64) * 100 - 50).reshape(P, 2).contiguous()
        del centre
        pt_off = (torch.arange(B + 1, device=dev, dtype=torch.int64) * ppb).to(torch.int32)
        out_box = torch.empty((B, 4), dtype=torch.float64, device=dev); out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
        out_high = torch.empty(N, dtype=torch.uint8, device=dev)
        alg = 16 * P + 4 * (B + 1) + 48 * B + 4 * (N + 1) + N
        res, chk = {}, {}
        for variant in (4, 6, 1, -1, 4, 6, 1, -1):
            _native.check(L.dyd_set_option(b"fused_variant", variant), "opt")
            ts = []
            for it in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                _native.check(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, P, 2, 0.98, out_box.data_ptr(),
                                                       out_arg.data_ptr(), out_high.data_ptr(), sp), "k12")
                b.record(); b.synchronize()
                if it >= 2:
                    ts.append(a.elapsed_time(b))
            res.setdefault(variant, []).append(float(np.median(ts)))
            chk[variant] = (int(out_high.sum().item()), float(out_box.sum().item()), int(out_arg.sum().item()))
        _native.check(L.dyd_set_option(b"fused_variant", -1), "opt")
        best = {v: min(t) for v, t in res.items()}
        print(json.dumps({"points_per_box": ppb, "rows": N, "boxes": B, "alg_GB": round(alg / 1e9, 3),
                          "wave_ms": round(best[4], 4), "workgroup_ms": round(best[6], 4), "two_launches_ms": round(best[1], 4), "auto_ms": round(best[-1], 4),
                          "auto_TBs": round(alg / best[-1] / 1e9, 2), "best_TBs": round(alg / min(best.values()) / 1e9, 2),
                          "same_results": len(set(chk.values())) == 1}), flush=True)
        del xy, pt_off, box_off, out_box, out_arg, out_high, nb


if __name__ == "__main__":
    main()
