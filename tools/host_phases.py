#!/usr/bin/env python3
"""Wall-clock of every host phase of the replace + IoU steps on a synthetic JSON table (no GPU needed).

    python tools/host_phases.py --rows 50000 [--threads 8]

The device stage is left out (K1's arg indices come from the C oracle, measurement aid only), so the numbers
are the host floor of the DataFrame -> DataFrame path: UTF-8 views of the str cells + native scan, native emit,
utf-8 -> str objects.
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()

    from deal_yolo_daya_amd import native_json as nj, synth
    from oracle import lib as olib

    t0 = time.perf_counter()
    t = synth.generate(args.rows, seed=synth.SEED)
    cells = synth.json_cells(t) if hasattr(synth, "json_cells") else [synth.row_json(t, r) for r in range(t.n_rows)]
    print(f"generate {args.rows} rows: {time.perf_counter() - t0:.2f}s, {sum(map(len, cells)) / 1e6:.1f} MB of JSON")

    for rep in range(args.repeat):
        ph = {}
        a = time.perf_counter()
        scan = nj.scan_polygons(cells, args.threads)
        ph["views+scan"] = time.perf_counter() - a
        a = time.perf_counter()
        _, arg4 = olib.bbox_minmax(scan.xy, scan.pt_off)
        ph["(oracle K1)"] = time.perf_counter() - a
        a = time.perf_counter()
        text, toff = scan.emit_buffers(arg4, args.threads)
        ph["emit"] = time.perf_counter() - a
        a = time.perf_counter()
        out = scan.emit(arg4, args.threads)
        ph["emit+str objects"] = time.perf_counter() - a
        host = ph["views+scan"] + ph["emit+str objects"]
        print(f"rep {rep}: " + "  ".join(f"{k} {v * 1e3:.0f} ms" for k, v in ph.items())
              + f"  | host floor {args.rows / host / 1e3:.1f} k rows/s; out {len(text) / 1e6:.0f} MB")
        scan.close()
        del out


if __name__ == "__main__":
    main()
