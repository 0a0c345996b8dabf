#!/usr/bin/env python3
"""Backs bench.py's `cpu_baseline.kind == "port"`: wall-clock of oracle/steps.py (the CPU restatement bench.py times on the GPU
box, where the reference cannot travel) against the reference itself, imported from /root/reference — BUILD CONTAINER ONLY.

Same 10^4-row synthetic CSV, the replace step then the IoU step, path in / path out, three alternating repetitions; SURVEY §8d
asks for agreement within +-10 %.  Writes profiles/cpu_port_vs_reference.json.

    python tools/cpu_port_vs_reference.py [--rows 10000]
"""
import argparse
import contextlib
import io
import json
import os
import statistics
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()

    import src.deal_yolo_data.core.processor as ref
    from deal_yolo_daya_amd import synth
    from oracle import steps as osteps

    df = synth.to_frame(synth.generate(args.rows, seed=synth.SEED))
    out = {"rows": args.rows, "runs": []}
    with tempfile.TemporaryDirectory() as d:
        inp = os.path.join(d, "in.csv")
        df.to_csv(inp, index=False, encoding="utf-8-sig")
        out["csv_mb"] = round(os.path.getsize(inp) / 1e6, 1)

        def run(kind):
            p, x, hi, lo = (os.path.join(d, f"{kind}_{n}.csv") for n in ("p", "x", "hi", "lo"))
            with contextlib.redirect_stdout(io.StringIO()):
                t0 = time.perf_counter()
                if kind == "reference":
                    ref.process_csv_replace_ptlist(inp, p, x)
                    t1 = time.perf_counter()
                    ref.filter_by_box_count_and_iou(p, hi, lo, 2, 0.98)
                else:
                    osteps.replace_csv(inp, p, x)
                    t1 = time.perf_counter()
                    osteps.iou_filter_csv(p, hi, lo, 2, 0.98)
                t2 = time.perf_counter()
            return {"kind": kind, "replace_s": round(t1 - t0, 3), "iou_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3)}, (p, hi, lo)

        files = {}
        for _ in range(args.reps):
            for kind in ("reference", "port"):
                rec, fs = run(kind)
                out["runs"].append(rec)
                files[kind] = fs
        same = all(open(a, "rb").read() == open(b, "rb").read() for a, b in zip(files["reference"], files["port"]))
    med = {k: statistics.median(r["total_s"] for r in out["runs"] if r["kind"] == k) for k in ("reference", "port")}
    out["median_total_s"] = med
    out["port_over_reference"] = round(med["port"] / med["reference"], 3)
    out["rows_per_s"] = {k: round(args.rows / v, 1) for k, v in med.items()}
    out["outputs_identical"] = same
    out["host"] = f"build container, {os.cpu_count()} logical cores, 1 used"
    path = os.path.join(REPO, "profiles", "cpu_port_vs_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("rows", "median_total_s", "port_over_reference", "rows_per_s", "outputs_identical")}))


if __name__ == "__main__":
    main()
