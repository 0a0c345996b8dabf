#!/usr/bin/env python3
"""Host-inclusive timing of the split step at table scale (GPU box): a `--rows` table goes through replace_and_filter_frame,
its `other` frame through split_frames `--reps` times (object columns, then Arrow text columns), phases printed per run; then the
gathers of the step in isolation, threaded helper against numpy.  One JSON line per measurement."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--micro", type=int, default=1)
    args = ap.parse_args()
    import pandas as pd
    import torch
    from deal_yolo_daya_amd import pycells, synth
    from deal_yolo_daya_amd.core import processor as P

    dev = torch.device("cuda", 0)
    parts = []
    for ci, s in enumerate(range(0, args.rows, 500_000)):
        t = synth.table_from_device(synth.generate_device(min(500_000, args.rows - s), synth.SEED + 77 + ci, dev))
        parts.append(pd.DataFrame({"source": synth.urls(t), synth.ANN_COL: synth.json_cells(t)}))
    df = pd.concat(parts, ignore_index=True)
    del parts
    kept, excluded, high, other = P.replace_and_filter_frame(df, 2, 0.98)
    del df, kept, excluded, high
    rules = synth.rules()
    for dtype in ("object", "arrow"):
        for rep in range(args.reps):
            st = {}
            a = time.perf_counter()
            res = P.split_frames(other, rules, stats=st, text_dtype=dtype)
            dt = time.perf_counter() - a
            print(json.dumps({"what": "split_frames", "text_dtype": dtype, "rep": rep, "rows_in": len(other), "seconds": round(dt, 3),
                              "rows_per_s": round(len(other) / dt), **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}},
                             ensure_ascii=False), flush=True)
            a = time.perf_counter()
            del res
            print(json.dumps({"what": "release of the result", "seconds": round(time.perf_counter() - a, 3)}), flush=True)
    if not args.micro:
        return
    rng = np.random.default_rng(0)
    n, m = len(other), 4_000_000
    src_vals = other["source"].to_numpy()
    w = other["width"].to_numpy()
    idx = rng.integers(0, n, m)

    def T(name, fn, reps=3):
        best = 1e9
        for _ in range(reps):
            t = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t); del r
        print(json.dumps({"what": "micro", "name": name, "elements": m, "ms": round(best * 1e3, 2)}), flush=True)

    T("take object column (threads)", lambda: pycells.take(src_vals, idx, checked=True))
    T("numpy object column", lambda: src_vals[idx])
    T("take int64 column (threads)", lambda: pycells.take(w, idx, checked=True))
    T("numpy int64 column", lambda: w[idx])
    tab = np.array([f"c{i}" for i in range(20)], object)
    codes = rng.integers(0, 20, 2 * m).astype(np.int32)
    order = rng.permutation(2 * m)[:m]
    T("take_small (threads)", lambda: pycells.take_small(tab, codes, order))
    T("numpy small", lambda: tab[codes[order]])
    L = 164
    text = np.full(m * L, 97, np.uint8)
    ptr = (text.ctypes.data + np.arange(m, dtype=np.int64) * L).astype(np.uint64)
    lens = np.full(m, L, np.int64)
    perm = rng.permutation(m)
    T("strings_from_views 164 B, shuffled, ascii hint", lambda: pycells.strings_from_views(ptr, lens, perm, all_ascii=True))
    T("strings_from_views 164 B, shuffled, classify", lambda: pycells.strings_from_views(ptr, lens, perm))
    T("strings_from_views 164 B, 1 thread", lambda: pycells.strings_from_views(ptr, lens, perm, n_threads=1, all_ascii=True))
    T("gather_text 164 B, shuffled", lambda: pycells.gather_text(ptr, lens, perm))


if __name__ == "__main__":
    main()
