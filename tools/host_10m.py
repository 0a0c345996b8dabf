#!/usr/bin/env python3
"""The host-inclusive replace -> IoU pass at the size north_star quotes: a `--rows` (default 10 M) row DataFrame of annotation
cells (45 GB of JSON at 10 M) through replace_and_filter_frame once; rows/s, phases and the process's peak RSS as one JSON line."""
import argparse
import json
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--split", type=int, default=0, help="1 = also run split_frames on the other frame (Arrow text columns)")
    args = ap.parse_args()
    import pandas as pd
    import torch
    from deal_yolo_daya_amd import _native, synth
    from deal_yolo_daya_amd.core import processor as P

    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    cells = np.empty(args.rows, object)
    for ci, s in enumerate(range(0, args.rows, 500_000)):
        n = min(500_000, args.rows - s)
        t = synth.table_from_device(synth.generate_device(n, synth.SEED + 77 + ci, dev))
        cells[s:s + n] = synth.json_cells(t)
        del t
        if ci % 4 == 3:
            print(f"generated {s + n} rows, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    torch.cuda.empty_cache()
    ids = np.random.default_rng(5).integers(0, int(0.9 * args.rows) + 1, args.rows)
    src = np.empty(args.rows, object)
    src[:] = [f"http://img.example/{k}.jpg" for k in ids.tolist()]
    df = pd.DataFrame({"source": src, synth.ANN_COL: cells}, copy=False)
    del cells, src, ids
    gen_s = time.perf_counter() - t0
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    stats = {}
    a = time.perf_counter()
    kept, excluded, high, other = P.replace_and_filter_frame(df, 2, 0.98, stats=stats)
    dt = time.perf_counter() - a
    out = {"what": "replace_and_filter_frame, DataFrame in -> frames out, one run", "rows": args.rows, "seconds": round(dt, 3),
           "rows_per_s": round(args.rows / dt), "high_rows": int(len(high)), "boxes": int(stats.get("boxes", 0)), "points": int(stats.get("points", 0)),
           "fused_launches": int(stats.get("fused_launches", 0)), "phases_s": {k[2:]: round(v, 3) for k, v in stats.items() if k.startswith("s_")},
           "table_generation_s": round(gen_s, 1), "rss_before_gb": round(rss0, 1),
           "peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1), "device": _native.device_name()}
    if args.split:
        del kept, excluded, high, df
        st = {}
        a = time.perf_counter()
        res = P.split_frames(other, synth.rules(), stats=st, text_dtype="arrow")
        d2 = time.perf_counter() - a
        out["split_frames_arrow"] = {"rows_in": int(len(other)), "seconds": round(d2, 3), "rows_per_s": round(len(other) / d2), "records": st.get("records"),
                                     "peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}
    print(json.dumps(out, ensure_ascii=False), flush=True)


if __name__ == "__main__":
    main()
