#!/usr/bin/env python3
"""Turn gpurun_out/ (rocprofv3 CSVs from tools/gpu_profile.sh) into the committed summaries:
profiles/<tag>_bench_kernel_stats.csv, profiles/<tag>_pmc_summary.json, profiles/<tag>_bench.json,
profiles/<tag>_bench_under_rocprof.json (bench.py's line from the run under the kernel trace).
usage: python tools/collect_profiles.py r01"""
import glob
import json
import os
import sys

import pandas as pd

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)

stats = glob.glob(os.path.join(out, "prof_bench", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    df = pd.read_csv(max(stats, key=os.path.getmtime))
    df = df[df["Name"].str.contains("dyd::")]
    df.to_csv(os.path.join(prof, f"{tag}_bench_kernel_stats.csv"), index=False)
    print(df[["Name", "Calls", "AverageNs", "Percentage"]].to_string())

summary = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:   # gpurun merges every call's files into the same directory: the latest run only
        df = pd.read_csv(f)
        df = df[df["Kernel_Name"].str.contains("dyd::")]
        for (k, c), g in df.groupby(["Kernel_Name", "Counter_Name"]):
            short = k.split("(")[0].replace("void ", "")
            summary.setdefault(short, {})[c] = {"mean_per_launch": float(g["Counter_Value"].mean()), "launches": int(len(g))}
for k, v in summary.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        # MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half
        # of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B stores.
        f, w = v["FETCH_SIZE"]["mean_per_launch"], v["WRITE_SIZE"]["mean_per_launch"]
        v["hbm_traffic_bytes_per_launch"] = {"read_corrected_x2": 2 * f * 1024, "read_raw": f * 1024, "write": w * 1024,
                                             "total_corrected": (2 * f + w) * 1024}
with open(os.path.join(prof, f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({k: v.get("hbm_traffic_bytes_per_launch") for k, v in summary.items()}, indent=1))

bl = os.path.join(out, "bench.log")
if os.path.exists(bl):
    lines = [l for l in open(bl) if l.startswith("{")]
    if lines:
        with open(os.path.join(prof, f"{tag}_bench.json"), "w") as fh:
            fh.write(lines[-1])

rl = os.path.join(out, "rocprof_bench.log")                 # the same command's own line while the kernel trace was on
if os.path.exists(rl):
    lines = [l for l in open(rl, errors="replace") if l.startswith("{")]
    if lines:
        with open(os.path.join(prof, f"{tag}_bench_under_rocprof.json"), "w") as fh:
            fh.write(lines[-1])

bj = os.path.join(prof, f"{tag}_bench.json")
if os.path.exists(bj):
    bench = json.load(open(bj))
    cfg = bench["config"]
    key = next((k for k in summary if "k12_wave_kernel" in k), None) or next((k for k in summary if "k12_" in k), None)
    t = summary.get(key, {}).get("hbm_traffic_bytes_per_launch") if key else None
    workload = "c3" if cfg["rows_per_gpu"] == 10_000_000 else ("c2" if cfg["rows_per_gpu"] == 1_000_000 else "other")
    if t:
        with open(os.path.join(prof, "k12_traffic.json"), "w") as fh:
            json.dump({"workload": workload, "kernel": key, "rows_per_gpu": cfg["rows_per_gpu"],
                       "traffic_bytes_per_launch": t["total_corrected"], "read_bytes_fetch_size_x2": t["read_corrected_x2"],
                       "write_bytes": t["write"],
                       "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
                       "traffic_over_algorithmic": t["total_corrected"] / bench["roofline"]["algorithmic_bytes_per_launch"],
                       "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of "
                                 "bench.py --steps 3 --warmup 1 on the same table)"},
                      fh, indent=1)

    dense = bench.get("dense")
    dkey = next((k for k in summary if "k12_wave_dense_kernel" in k), None)
    dt = summary.get(dkey, {}).get("hbm_traffic_bytes_per_launch") if dkey else None
    if dense and dt:
        alg = dense["roofline"]["algorithmic_bytes_per_launch"]
        with open(os.path.join(prof, f"{tag}_c5_traffic.json"), "w") as fh:
            json.dump({"workload": "c5", "kernel": dkey, "rows_per_gpu": dense["rows"], "traffic_bytes_per_launch": dt["total_corrected"],
                       "read_bytes_fetch_size_x2": dt["read_corrected_x2"], "write_bytes": dt["write"], "algorithmic_bytes_per_launch": alg,
                       "traffic_over_algorithmic": dt["total_corrected"] / alg,
                       "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of "
                                 "bench.py --workload c5 --steps 3 --warmup 1: the table the dense object of the default line draws)"},
                      fh, indent=1)
