#!/usr/bin/env python3
"""One image with very many boxes among ordinary ones: K2 through the host entry (rows above 1024 boxes go to the big-row
kernel) and the fused device entry (which keeps such a row on one wave).
    python tools/bigrow_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from deal_yolo_daya_amd import _native
    L = _native.lib()
    dev = torch.device("cuda:0")
    sp = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(1)
    for big in (0, 1000, 10000, 50000):
        nb = torch.randint(1, 33, (100000,), generator=g, device=dev)
        if big:
            nb[50000] = big
        box_off = torch.zeros(nb.numel() + 1, dtype=torch.int32, device=dev)
        box_off[1:] = torch.cumsum(nb, 0).to(torch.int32)
        N, B = nb.numel(), int(box_off[-1].item())
        ppb = 8
        P = B * ppb
        centre = torch.rand((B, 1, 2), generator=g, device=dev, dtype=torch.float64) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)
        xy = (centre + torch.rand((B, ppb, 2), generator=g, device=dev, dtype=torch.float64) * 100 - 50).reshape(P, 2).contiguous()
        pt_off = (torch.arange(B + 1, device=dev, dtype=torch.int64) * ppb).to(torch.int32)
        ob = torch.empty((B, 4), dtype=torch.float64, device=dev); oa = torch.empty((B, 4), dtype=torch.int32, device=dev)
        oh = torch.empty(N, dtype=torch.uint8, device=dev)
        ts = []
        for it in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _native.check(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, P, 2, 0.98, ob.data_ptr(),
                                                   oa.data_ptr(), oh.data_ptr(), sp), "k12")
            b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        box_h, off_h = ob.cpu().numpy(), box_off.cpu().numpy()
        _native.iou_any_ge(box_h, off_h, 2, 0.98)
        t0 = time.perf_counter()
        high = _native.iou_any_ge(box_h, off_h, 2, 0.98)
        host_s = time.perf_counter() - t0
        print(json.dumps({"rows": N, "boxes": B, "one_row_of": big, "fused_dev_ms": round(float(np.median(ts[1:])), 3),
                          "k2_host_entry_kernels_ms": round(_native.last_kernel_ms(), 3), "k2_host_entry_call_ms": round(host_s * 1e3, 2),
                          "same_flags": bool(np.array_equal(high, oh.cpu().numpy()))}), flush=True)


if __name__ == "__main__":
    main()
