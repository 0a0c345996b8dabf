#!/usr/bin/env python3
"""Host-inclusive throughput of the replace + IoU steps, DataFrame in -> frames out (SURVEY §8d region 2): one native scan,
one fused K1+K2 launch, one native emit.  Prints one JSON line per repetition with the phase split.

    python tools/e2e_bench.py --rows 1000000 --reps 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", type=int, default=2000, help="rows compared with the CPU port of the reference (0 = none)")
    ap.add_argument("--csv", type=int, default=0, help="1 = also time the path-in / path-out route (SURVEY §8d region 3): the fused CSV "
                                                       "twin and the two step functions in sequence, files under --dir")
    ap.add_argument("--dir", default="/tmp/dyd_e2e")
    ap.add_argument("--dedup-rows", type=int, default=0, help="also time the in-memory dedup / reference filter on this many URL rows")
    args = ap.parse_args()

    import pandas as pd
    from deal_yolo_daya_amd import _native, native_json, synth
    from deal_yolo_daya_amd.core import processor as P

    _native.lib()
    if args.dedup_rows:
        n = args.dedup_rows
        rng = np.random.default_rng(5)
        ids = rng.integers(0, int(0.9 * n) + 1, size=n)
        src = pd.Series(np.char.add(np.char.add("http://img.example/", ids.astype(str)), ".jpg").astype(object), name="source")
        src.iloc[::1000003] = np.nan
        ref = pd.Series([f"http://img.example/{k}.jpg" for k in range(0, int(0.9 * n) + 1, 10)], name="source")
        for rep in range(2):
            a = time.perf_counter(); m1 = P.dedup_keep_mask(src, "first"); b = time.perf_counter()
            w1 = ~src.duplicated(keep="first").to_numpy(); c = time.perf_counter()
            m2 = P.ref_hit_mask(src, ref); d = time.perf_counter()
            w2 = src.astype(str).isin(set(ref.dropna().astype(str))).to_numpy(); e = time.perf_counter()
            print(json.dumps({"dedup_rows": n, "dedup_keep_mask_s": round(b - a, 3), "pandas_duplicated_s": round(c - b, 3),
                              "ref_hit_mask_s": round(d - c, 3), "pandas_isin_s": round(e - d, 3),
                              "identical": bool(np.array_equal(m1, w1) and np.array_equal(m2, w2)), "kernel_ms": round(_native.last_kernel_ms(), 3)}),
                  flush=True)
        if args.rows <= 0:
            return
    t0 = time.perf_counter()
    parts = []
    for ci, s in enumerate(range(0, args.rows, 250_000)):
        t = synth.generate(min(250_000, args.rows - s), seed=synth.SEED + ci)
        parts.append(pd.DataFrame({"source": synth.urls(t), synth.ANN_COL: synth.json_cells(t)}))
        del t
    df = pd.concat(parts, ignore_index=True)
    del parts
    n_bytes = int(df[synth.ANN_COL].str.len().sum())
    print(f"# table: {len(df)} rows, {n_bytes / 1e9:.2f} GB of JSON, generated in {time.perf_counter() - t0:.0f}s; "
          f"host threads {native_json.host_threads()}", file=sys.stderr, flush=True)
    for rep in range(args.reps):
        stats = {}
        a = time.perf_counter()
        kept, excluded, high, other = P.replace_and_filter_frame(df, 2, 0.98, stats=stats)
        dt = time.perf_counter() - a
        out_bytes = int(kept[P.BBOX_COL].str.len().sum()) if rep == 0 else None
        print(json.dumps({"rows": len(df), "seconds": round(dt, 3), "rows_per_s": round(len(df) / dt), "high": len(high),
                          "phases_s": {k: round(v, 3) for k, v in stats.items() if k.startswith("s_")},
                          "fast_cells": stats.get("fast_cells"), "python_cells": stats.get("python_cells"),
                          "out_bytes": out_bytes, "kernel_ms": round(_native.last_kernel_ms(), 3)}), flush=True)
    for rep in range(2):      # the same with the bbox column as pandas' Arrow-backed string dtype (no str objects)
        stats = {}
        a = time.perf_counter()
        kept, excluded, high, other = P.replace_and_filter_frame(df, 2, 0.98, stats=stats, text_dtype="arrow")
        dt = time.perf_counter() - a
        print(json.dumps({"text_dtype": "arrow", "bbox_dtype": str(kept[P.BBOX_COL].dtype), "rows": len(df), "seconds": round(dt, 3),
                          "rows_per_s": round(len(df) / dt), "high": len(high),
                          "phases_s": {k: round(v, 3) for k, v in stats.items() if k.startswith("s_")}}), flush=True)
        del kept, excluded, high, other
    if args.csv:
        os.makedirs(args.dir, exist_ok=True)
        Q = lambda n: os.path.join(args.dir, n)  # noqa: E731
        a = time.perf_counter()
        df.to_csv(Q("in.csv"), index=False, encoding="utf-8-sig")
        print(json.dumps({"csv": "pandas to_csv of the input table", "seconds": round(time.perf_counter() - a, 2),
                          "bytes": os.path.getsize(Q("in.csv"))}), flush=True)
        import contextlib
        import io
        from deal_yolo_daya_amd import fastcsv as _fc
        phase = {}

        def timed(name, fn):
            def w(*a, **k):
                t = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    phase[name] = phase.get(name, 0.0) + time.perf_counter() - t
            return w
        _fc.read_split = timed("read_split", _fc.read_split)
        _fc.write_table = timed("write_table", _fc.write_table)
        P._replace_csv_core = timed("core(read+scan+launch+emit)", P._replace_csv_core)
        P._as_reread = timed("as_reread", P._as_reread)
        for rep in range(2):
            phase.clear()
            with contextlib.redirect_stdout(io.StringIO()):
                a = time.perf_counter()
                P.process_csv_replace_and_filter(Q("in.csv"), Q("p.csv"), Q("x.csv"), Q("hi.csv"), Q("lo.csv"), 2, 0.98)
                t_fused = time.perf_counter() - a
                a = time.perf_counter()
                P.process_csv_replace_ptlist(Q("in.csv"), Q("p2.csv"), Q("x2.csv"))
                b = time.perf_counter()
                P.filter_by_box_count_and_iou(Q("p2.csv"), Q("hi2.csv"), Q("lo2.csv"), 2, 0.98)
                t_two = (b - a, time.perf_counter() - b)
            same = all(open(Q(x), "rb").read() == open(Q(y), "rb").read() for x, y in (("p.csv", "p2.csv"), ("hi.csv", "hi2.csv"), ("lo.csv", "lo2.csv")))
            print(json.dumps({"csv": "CSV -> CSV", "rows": len(df), "fused_twin_s": round(t_fused, 3), "fused_rows_per_s": round(len(df) / t_fused),
                              "replace_step_s": round(t_two[0], 3), "iou_step_s": round(t_two[1], 3),
                              "two_steps_rows_per_s": round(len(df) / sum(t_two)), "io_path": dict(P.LAST_IO_PATH),
                              "phases_fused_plus_two_steps_s": {k: round(v, 3) for k, v in phase.items()},
                              "same_files": same, "out_bytes": sum(os.path.getsize(Q(n)) for n in ("p.csv", "hi.csv", "lo.csv"))}), flush=True)
        for n in os.listdir(args.dir):
            os.remove(os.path.join(args.dir, n))
    if args.check:
        from oracle import steps as osteps
        sub = df.iloc[:args.check]
        kept, _, high, other = P.replace_and_filter_frame(sub, 2, 0.98)
        okept, oproj, _ = osteps.replace_frame(sub)
        ohi, olo = osteps.iou_filter_frame(oproj, 2, 0.98)
        pd.testing.assert_frame_equal(kept, okept)
        assert high["source"].tolist() == ohi["source"].tolist() and other["source"].tolist() == olo["source"].tolist()
        print(f"# first {args.check} rows identical to the CPU port of the reference", file=sys.stderr)


if __name__ == "__main__":
    main()
