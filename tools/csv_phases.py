#!/usr/bin/env python3
"""Where the native CSV hand-off spends its time (host only): index / extract / project / pandas / write.
    python tools/csv_phases.py --rows 20000"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20000)
    a = ap.parse_args()
    import io
    import pandas as pd
    from deal_yolo_daya_amd import fastcsv, synth
    from deal_yolo_daya_amd.core import processor as P
    df = synth.to_frame(synth.generate(a.rows, seed=synth.SEED))
    out = {"rows": a.rows}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "in.csv")
        df.to_csv(path, index=False, encoding="utf-8-sig")
        out["MB"] = round(os.path.getsize(path) / 1e6, 1)

        def clock(fn, rep=3):
            best = 1e9
            for _ in range(rep):
                t0 = time.perf_counter()
                r = fn()
                best = min(best, time.perf_counter() - t0)
            return round(best, 4), r
        out["read_bytes_s"], raw = clock(lambda: open(path, "rb").read())
        buf = np.frombuffer(raw, np.uint8)[3:]
        out["index_s"], idx = clock(lambda: fastcsv.CsvIndex.open(buf))
        c = idx.names.index(P.ANNOTATION_COL)
        out["col_bytes_s"], _ = clock(lambda: idx.col_bytes(c))
        out["extract_s"], col = clock(lambda: idx.extract(c))
        keep = [i for i in range(len(idx.names)) if i != c]
        out["project_s"], text = clock(lambda: idx.project(keep))
        out["pandas_light_s"], light = clock(lambda: pd.read_csv(io.BytesIO(text), encoding="utf-8", usecols=[idx.names[i] for i in keep]))
        out["read_split_total_s"], t = clock(lambda: fastcsv.read_split(path, [P.ANNOTATION_COL]))
        cols = [t.heavy[n] if n in t.heavy else t.light[n] for n in t.names]
        out["write_table_s"], _ = clock(lambda: fastcsv.write_table(os.path.join(d, "o.csv"), t.names, cols, t.n_rows))
        out["frame_from_split_s"], _ = clock(lambda: fastcsv.frame_from_split(t))
        out["pandas_read_csv_s"], _ = clock(lambda: pd.read_csv(path, encoding="utf-8-sig"), rep=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
