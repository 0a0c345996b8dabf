#!/usr/bin/env python3
"""Phase timeline of K7 tiles (tuning aid): per-tile wall_clock64 stamps -> median phase durations.
    python tools/k7_trace.py [--rows 16000000] [--variant 2]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16_000_000)
    ap.add_argument("--variant", type=int, default=2, help="the phase stamps exist in the single-tile kernel (2) and the box-tiled kernel (30)")
    ap.add_argument("--boxes-per-row", type=int, default=1, help="variant 30: every row holds this many boxes")
    ap.add_argument("--measure", action="store_true")
    a = ap.parse_args()
    import torch
    from deal_yolo_daya_amd import _native
    L = _native.lib()
    dev = torch.device("cuda:0")
    n = a.rows
    g = torch.Generator(device=dev).manual_seed(1)
    c = torch.rand((n, 2), generator=g, device=dev, dtype=torch.float64) * 1000
    box = torch.round(torch.cat([c, c + torch.rand((n, 2), generator=g, device=dev, dtype=torch.float64) * 200 + 1], 1) * 100) / 100
    bpr = a.boxes_per_row
    n_boxes, n = n, n // bpr
    box = box[: n * bpr].contiguous()
    off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * bpr).to(torch.int32)
    w = torch.full((n,), 1920.0, dtype=torch.float64, device=dev); h = torch.full((n,), 1080.0, dtype=torch.float64, device=dev)
    cid = (torch.arange(n, device=dev, dtype=torch.int32) % 20).contiguous()
    toff = torch.empty(n + 1, dtype=torch.int64, device=dev); flag = torch.empty(n, dtype=torch.uint8, device=dev)
    text = torch.empty(44 * n * bpr, dtype=torch.uint8, device=dev)
    total = C.c_int64()
    _native.check(L.dyd_set_option(b"k7_variant", a.variant), "opt")
    tile = 256 * a.variant if a.variant != 30 else 480
    n_tiles = ((n if a.variant != 30 else n * bpr) + tile - 1) // tile
    trace = torch.zeros(n_tiles * 8, dtype=torch.int64, device=dev)

    def run():
        _native.check(L.dyd_yolo_lines_dev(box.data_ptr(), off.data_ptr(), None, w.data_ptr(), h.data_ptr(), cid.data_ptr(), n, n * bpr, toff.data_ptr(),
                                           flag.data_ptr(), None if a.measure else text.data_ptr(), text.numel(), C.byref(total), None), "k7")
    run(); run()
    _native.check(L.dyd_set_option(b"k7_trace_ptr", trace.data_ptr()), "opt")
    run()
    _native.check(L.dyd_set_option(b"k7_trace_ptr", 0), "opt")
    t = trace.cpu().numpy().reshape(n_tiles, 8).astype(np.float64) / 100.0        # wall_clock64 ticks at 100 MHz -> microseconds
    names = ["load_issue->loaded", "measure", "scan", "print", "lookback", "flush"]
    last = 5 if a.measure else 6
    if a.variant == 30:
        names = ["rows+boxes loaded", "search+convert", "scans", "print to LDS", "lookback", "row outputs", "flush"]
        last = 6 if a.measure else 7
    out = {"rows": n, "variant": a.variant, "tiles": n_tiles, "kernel_span_us": float(t[:, :last + 1].max() - t[:, 0].min())}
    for i in range(last):
        d = t[:, i + 1] - t[:, i]
        out[names[i]] = {"median_us": float(np.median(d)), "p90_us": float(np.percentile(d, 90))}
    life = t[:, last] - t[:, 0]
    out["tile_lifetime_us"] = {"median": float(np.median(life)), "p90": float(np.percentile(life, 90))}
    start = np.sort(t[:, 0]) - t[:, 0].min()
    out["tile_starts_per_us"] = float(n_tiles / (start[-1] + 1e-9))
    if a.variant != 30:
        out["plain_flag_counts"] = {int(k): int(v) for k, v in zip(*np.unique(trace.cpu().numpy().reshape(n_tiles, 8)[:, 7], return_counts=True))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
