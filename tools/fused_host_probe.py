#!/usr/bin/env python3
"""Where does the host-pointer fused entry (dyd_bbox_iou_fused: alloc, H2D, launch, D2H, free) spend its time — one call alone, then 16 calls side by
side as the native replace -> IoU pipeline issues them.  DYD_TRACE_FUSED=1 makes the library print its own phase split per call.
    python tools/fused_host_probe.py [--rows 62500] [--threads 16]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=62500)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    from deal_yolo_daya_amd import _native, synth
    _native.lib()
    t = synth.generate(a.rows, seed=3)
    xy, po, bo = np.ascontiguousarray(t.xy), t.pt_off, t.box_off
    mb = xy.nbytes / 1e6
    for _ in range(2):
        _native.bbox_iou_fused(xy, po, bo, 2, 0.98)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); _native.bbox_iou_fused(xy, po, bo, 2, 0.98); ts.append(time.perf_counter() - t0)
    out = {"rows": a.rows, "points_MB": round(mb, 1), "alone_s": round(min(ts), 4), "alone_GBs": round(mb / 1e3 / min(ts), 2)}
    copies = [xy.copy() for _ in range(a.threads)]
    res = [0.0] * a.threads

    def work(k):
        t0 = time.perf_counter(); _native.bbox_iou_fused(copies[k], po, bo, 2, 0.98); res[k] = time.perf_counter() - t0
    for rep in range(3):
        th = [threading.Thread(target=work, args=(k,)) for k in range(a.threads)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        wall = time.perf_counter() - t0
        out[f"side_by_side_{rep}"] = {"wall_s": round(wall, 4), "slowest_call_s": round(max(res), 4), "fastest_call_s": round(min(res), 4),
                                      "aggregate_GBs": round(a.threads * mb / 1e3 / wall, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
