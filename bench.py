#!/usr/bin/env python3
"""bench.py — annotation rows/s through the poly->bbox + IoU-filter path on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (K1 polygon->bbox and K2 all-pairs IoU flag in ONE fused launch,
dyd_bbox_iou_fused_dev — the launch the product's replace -> IoU step functions issue) over one batch of synthetic rows
that is already resident in HBM.  Default workload = the configuration BASELINE.json's target is quoted on: 10M rows per
GPU (configs[2]'s table: <=32 boxes/image, 3..12 points/polygon, 1.24 G points = 28 GB with outputs); weak scaling
(every rank owns its own shard, rows are independent: no data-path collective).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline       dominant kernel (the fused K1+K2 kernel) — algorithmic bytes / HIP-event time vs 8 TB/s HBM peak
  cpu_baseline   the CPU port of the reference path (oracle/steps.py) timed on this box's host cores (rank 0, N=1 only)
  host_inclusive SURVEY §8d region 2: DataFrame in -> frames out through the product's step function
                 (replace_and_filter_frame: scan + H2D + fused launch + D2H + emit) and its ratio to cpu_baseline;
                 .pipeline = configs[2] through the product API: the five step functions in sequence on that table, per step
                 seconds and the ratio to the same step of the CPU port
                 .path_io = SURVEY §8d region 3: CSV path in -> CSV paths out, the fused twin and the page's two step functions
  full_pipeline  configs[2]: K3 -> K4 -> K5 -> K1+K2 -> permutation + K6 on the same 10M resident rows, per stage
  dense          configs[4] scaled (1M rows x 256 boxes, 44 GB): the same launch, its own roofline object
and the sharded dedup / reference filter of configs[3] with its collectives timed on their own: sharded_exchange (weak: rows per GPU
fixed, at N > 1 or --workload c4) and sharded_exchange_strong (a fixed 100 M-row table cut into N shards, at every N incl. 1); at N > 1
also fused_strong: the fused launch on every rank's shard of that fixed table (configs[3] itself at N = 8).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
MIN_BOXES, THR = 2, 0.98

WORKLOADS = {
    # name: (rows per GPU, boxes_per_row or None, description)
    "c2": (1_000_000, None, "configs[1]: 1M rows ptList->bbox + IoU filter (<=32 boxes/img) per GPU"),
    "c3": (10_000_000, None, "configs[2] table: 10M rows (<=32 boxes/img), ptList->bbox + IoU filter per GPU, inputs resident in HBM"),
    "c4": (12_500_000, None, "configs[3]: 100M rows over 8 GPUs = 12.5M rows per GPU (<=32 boxes/img): ptList->bbox + IoU filter per rank, "
                             "then the sharded URL dedup / reference filter with its all-gathers (sharded_exchange)"),
    "c5": (1_000_000, 256, "configs[4] scaled: dense-box stress, 256 boxes/img, polygons of 3..12 points"),
}
GEN_CHUNK = 2_000_000


def cpu_baseline(sizes):
    """Time the reference algorithm's CPU port (oracle/steps.py, 1 core) on synthetic rows of the same generator: the replace
    step then the IoU step on the in-memory DataFrame (json.loads / dumps included, CSV I/O excluded)."""
    from deal_yolo_daya_amd import synth
    from oracle import steps as osteps

    runs = []
    for n in sizes:
        df = synth.to_frame(synth.generate(n, seed=synth.SEED))
        t0 = time.perf_counter()
        kept, projected, _ = osteps.replace_frame(df)
        t1 = time.perf_counter()
        high, other = osteps.iou_filter_frame(projected, MIN_BOXES, THR)
        t2 = time.perf_counter()
        runs.append({"rows": n, "rows_per_s": n / (t2 - t0), "replace_s": round(t1 - t0, 2), "iou_s": round(t2 - t1, 2),
                     "high_rows": int(len(high))})
        del df, kept, projected, high, other
    big = runs[-1]
    return {
        "value": big["rows_per_s"], "unit": "rows/s", "cores": 1, "kind": "port",
        "sample": (f"{big['rows']} synthetic rows (same generator as the GPU batch), in-memory DataFrame: replace step "
                   f"{big['replace_s']}s + IoU step {big['iou_s']}s, json.loads/dumps included, CSV I/O excluded; host has "
                   f"{os.cpu_count()} logical cores, 1 used"),
        "runs": runs,
    }


def host_table(rows, dev):
    """the DataFrame of the host-inclusive legs: `rows` annotation cells drawn on the device and emitted natively, URL ids ~
    U{0..0.9 rows} (about 40 % duplicates), and the reference frame of every id divisible by 10"""
    import pandas as pd
    import torch
    from deal_yolo_daya_amd import synth

    t0 = time.perf_counter()
    parts = []
    for ci, s in enumerate(range(0, rows, 500_000)):
        d = synth.generate_device(min(500_000, rows - s), synth.SEED + 77 + ci, dev)
        t = synth.table_from_device(d)
        del d
        parts.append(synth.json_cells(t))
        del t
    cells = np.concatenate(parts)
    del parts
    ids = np.random.default_rng(synth.SEED + 5).integers(0, int(0.9 * rows) + 1, rows)
    src = np.empty(rows, object)
    src[:] = [f"http://img.example/{k}.jpg" for k in ids.tolist()]
    df = pd.DataFrame({"source": src, synth.ANN_COL: cells})
    ref = pd.DataFrame({"source": synth.reference_urls(rows)})
    torch.cuda.empty_cache()
    return df, ref, time.perf_counter() - t0


def host_inclusive(df, gen_s):
    """DataFrame in -> (kept, excluded, high, other) frames out through the product's fused step function."""
    from deal_yolo_daya_amd import _native, native_json, synth
    from deal_yolo_daya_amd.core import processor as P

    rows = len(df)
    best, stats_best, n_high = None, None, 0
    for _ in range(2):
        stats = {}
        a = time.perf_counter()
        kept, excluded, high, other = P.replace_and_filter_frame(df, MIN_BOXES, THR, stats=stats)
        dt = time.perf_counter() - a
        if best is None or dt < best:
            best, stats_best, n_high = dt, stats, len(high)
        del kept, excluded, high, other
    json_bytes = int(df[synth.ANN_COL].str.len().sum())
    return {
        "region": "SURVEY §8d (2): DataFrame in -> kept / excluded / high / other frames out, replace_and_filter_frame "
                  "(UTF-8 views of the str cells, native scan, H2D, ONE fused K1+K2 launch, D2H, native emit, str objects); best of 2",
        "rows": rows, "seconds": best, "value": rows / best, "unit": "rows/s", "high_rows": n_high,
        "json_gb": round(json_bytes / 1e9, 2), "host_threads": native_json.host_threads(),
        "phases_s": {k[2:]: round(v, 3) for k, v in stats_best.items() if k.startswith("s_")},
        "fast_lane_cells": stats_best.get("fast_cells"), "python_cells": stats_best.get("python_cells"),
        "table_generation_s": round(gen_s, 1), "kernel_ms": round(_native.last_kernel_ms(), 3),
    }


def host_pipeline(df, ref, cpu_rows):
    """configs[2] through the product's own API: the five step functions in sequence on ONE table, each handing its frame to the
    next as the processing page does (reference ui/pages/processing.py:545-630; the replace and IoU steps as the one fused pass),
    wall-clock per step, ONE run (no best-of).  Beside every step: the same step of the CPU port (oracle/steps.py, 1 core) on the first
    `cpu_rows` rows of the same table."""
    from deal_yolo_daya_amd import synth
    from deal_yolo_daya_amd.core import processor as P
    from oracle import steps as osteps

    rules = synth.rules()
    steps = []

    def run(name, rows_in, fn):
        a = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - a
        steps.append({"step": name, "rows_in": int(rows_in), "seconds": round(dt, 4), "rows_per_s": round(rows_in / dt, 1)})
        return out

    t0 = time.perf_counter()
    dd = run("dedup_frame (K3 + K4)", len(df), lambda: P.dedup_frame(df))
    ff = run("ref_filter_frame (K3 + K5)", len(dd), lambda: P.ref_filter_frame(dd, ref))
    rstats, sstats = {}, {}
    kept, excluded, high, other = run("replace_and_filter_frame (scan, fused K1+K2, emit)", len(ff),
                                      lambda: P.replace_and_filter_frame(ff, MIN_BOXES, THR, stats=rstats))
    res = run("split_frames (native expansion, K8 + K6, frames)", len(other), lambda: P.split_frames(other, rules, stats=sstats))
    total = time.perf_counter() - t0
    steps[-1]["phases_s"] = {k: round(v, 3) for k, v in sstats.items() if k.endswith("_s") and not isinstance(v, dict)}
    steps[-1]["records"] = sstats.get("records")
    steps[-1]["category_counts"] = res["category_counts"]
    steps[-1]["unclassified"] = int(len(res["unclassified"]))
    steps[2]["high_rows"] = int(len(high))
    rows_out = {"dedup": len(dd), "ref_filter": len(ff), "kept": len(kept), "other": len(other)}
    del dd, ff, kept, excluded, high, other, res

    cpu = None
    if cpu_rows > 0:
        sub = df.iloc[:cpu_rows].reset_index(drop=True)
        cpu = []

        def crun(name, rows_in, fn):
            a = time.perf_counter()
            out = fn()
            cpu.append({"step": name, "rows_in": int(rows_in), "seconds": round(time.perf_counter() - a, 3),
                        "rows_per_s": round(rows_in / (time.perf_counter() - a), 1)})
            return out

        # the two key steps are pandas one-liners: timed on the WHOLE table (a 20 000-row sample sits in cache and flatters them)
        c1 = crun("dedup", len(df), lambda: osteps.dedup_frame(df))
        c2 = crun("ref_filter", len(c1), lambda: osteps.ref_filter_frame(c1, ref))
        del c1, c2
        c1 = osteps.dedup_frame(sub)
        c2 = osteps.ref_filter_frame(c1, ref)
        a = time.perf_counter()
        _, projected, _ = osteps.replace_frame(c2)
        _, c_other = osteps.iou_filter_frame(projected, MIN_BOXES, THR)
        dt = time.perf_counter() - a
        cpu.append({"step": "replace + iou_filter", "rows_in": len(c2), "seconds": round(dt, 3), "rows_per_s": round(len(c2) / dt, 1)})
        c_split = c_other.iloc[:max(1, cpu_rows // 10)]           # the port's split walks ~150 rows a second (row.copy() per record)
        crun("split", len(c_split), lambda: osteps.split_frames(c_split, rules))
        for st, c in zip(steps, cpu):
            st["vs_cpu_port"] = round(st["rows_per_s"] / c["rows_per_s"], 1)
    return {"config": "configs[2] through the product API: dedup_frame -> ref_filter_frame -> replace_and_filter_frame -> split_frames "
                      "on one table, host-inclusive (DataFrame in, DataFrames out), one run",
            "rows": len(df), "seconds": round(total, 3), "rows_per_s": round(len(df) / total, 1), "rows_out": rows_out,
            "steps": steps, "cpu_port": cpu,
            "cpu_port_sample": f"dedup, ref_filter: the whole table; replace + iou_filter: first {cpu_rows} rows of it through the first two steps; "
                               f"split: the first {max(1, cpu_rows // 10)} rows of its input; 1 core" if cpu else None}


def path_io(df, rows):
    """SURVEY §8d region (3), path in -> path out: the first `rows` rows of the host table as a CSV, through the fused CSV twin
    (process_csv_replace_and_filter: five files from one pass) and through the two step functions as the unchanged processing
    page calls them (process_csv_replace_ptlist, then filter_by_box_count_and_iou on the file it wrote), one run each."""
    import contextlib
    import io
    import shutil
    import tempfile

    from deal_yolo_daya_amd import fastcsv as _fc
    from deal_yolo_daya_amd import synth
    from deal_yolo_daya_amd.core import processor as P

    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (32 << 30) else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="dyd_bench_", dir=base)
    Q = lambda n: os.path.join(d, n)  # noqa: E731
    try:
        sub = df.iloc[:rows].reset_index(drop=True)
        a = time.perf_counter()
        if not _fc.write_table(Q("in.csv"), list(sub.columns), [sub[c] for c in sub.columns], len(sub)):
            sub.to_csv(Q("in.csv"), index=False, encoding="utf-8-sig")
        write_in = time.perf_counter() - a
        in_bytes = os.path.getsize(Q("in.csv"))
        P.clear_step_cache()
        with contextlib.redirect_stdout(io.StringIO()):
            a = time.perf_counter()
            P.process_csv_replace_and_filter(Q("in.csv"), Q("p.csv"), Q("x.csv"), Q("hi.csv"), Q("lo.csv"), MIN_BOXES, THR)
            t_fused = time.perf_counter() - a
            how_fused = P.LAST_IO_PATH.get("replace_iou")
            a = time.perf_counter()
            P.process_csv_replace_ptlist(Q("in.csv"), Q("p2.csv"), Q("x2.csv"))
            b = time.perf_counter()
            P.filter_by_box_count_and_iou(Q("p2.csv"), Q("hi2.csv"), Q("lo2.csv"), MIN_BOXES, THR)
            c = time.perf_counter()
        same = all(open(Q(x), "rb").read() == open(Q(y), "rb").read() for x, y in (("p.csv", "p2.csv"), ("hi.csv", "hi2.csv"), ("lo.csv", "lo2.csv")))
        out_bytes = sum(os.path.getsize(Q(n)) for n in ("p.csv", "hi.csv", "lo.csv"))
        return {"region": "SURVEY §8d (3): CSV path in -> CSV paths out (processed, excluded, high, other)", "rows": rows, "filesystem": base,
                "input_csv_bytes": in_bytes, "output_csv_bytes": out_bytes, "input_written_s": round(write_in, 3),
                "fused_twin": {"function": "process_csv_replace_and_filter", "seconds": round(t_fused, 3), "rows_per_s": round(rows / t_fused), "io_path": how_fused},
                "two_steps": {"functions": "process_csv_replace_ptlist -> filter_by_box_count_and_iou (import swap only)",
                              "replace_s": round(b - a, 3), "iou_s": round(c - b, 3), "seconds": round(c - a, 3), "rows_per_s": round(rows / (c - a)),
                              "io_path": {"replace": P.LAST_IO_PATH.get("replace"), "iou": P.LAST_IO_PATH.get("iou")},
                              "vs_fused_twin": round((c - a) / t_fused, 3)},
                "same_files": bool(same)}
    finally:
        P.clear_step_cache()
        shutil.rmtree(d, ignore_errors=True)


def full_pipeline(tab, dev, L, ck, sp):
    """configs[2] on the resident table: every stage on the FULL table (the real pipeline hands later stages only the surviving
    rows, so the sum is an upper bound), HIP-event timed on the launch stream, median of 3."""
    import torch
    from deal_yolo_daya_amd import _native

    N, B, P = tab["N"], tab["B"], tab["P"]

    def timeit(fn, iters=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    prefix, suffix = b"http://img.example/", b".jpg"
    width = len(prefix) + 9 + len(suffix)

    def url_bytes(ids):
        n = ids.numel()
        out = torch.empty((n, width), dtype=torch.uint8, device=dev)
        out[:, :len(prefix)] = torch.tensor(list(prefix), dtype=torch.uint8, device=dev)
        out[:, len(prefix) + 9:] = torch.tensor(list(suffix), dtype=torch.uint8, device=dev)
        v = ids.clone()
        for k in range(8, -1, -1):
            out[:, len(prefix) + k] = (v % 10 + 48).to(torch.uint8)
            v //= 10
        return out.reshape(-1), torch.arange(n + 1, device=dev, dtype=torch.int64) * width

    g = torch.Generator(device=dev).manual_seed(11)
    ids = torch.randint(0, int(0.9 * N) + 1, (N,), generator=g, device=dev, dtype=torch.int64)
    data, off = url_bytes(ids)
    ref_ids = torch.arange(0, int(0.9 * N) + 1, 10, device=dev, dtype=torch.int64)
    rdata, roff = url_bytes(ref_ids)
    R = int(ref_ids.numel())
    U = int(torch.unique(ids).numel())
    h = torch.empty((N, 2), dtype=torch.int64, device=dev)
    hr = torch.empty((R, 2), dtype=torch.int64, device=dev)
    keep = torch.empty(N, dtype=torch.uint8, device=dev)
    hit = torch.empty(N, dtype=torch.uint8, device=dev)
    stages = []

    def stage(name, nbytes, ms, **kw):
        stages.append({"stage": name, "ms": round(ms, 3), "alg_GB": round(nbytes / 1e9, 3), "GBs": round(nbytes / ms / 1e6, 1), **kw})

    stage("K3 hash128(source)", int(data.numel()) + 8 * (N + 1) + 16 * N,
          timeit(lambda: ck(L.dyd_hash128_dev(data.data_ptr(), off.data_ptr(), N, h.data_ptr(), sp), "k3")))
    stage("K4 dedup keep=first", 16 * N + N + 48 * U,
          timeit(lambda: ck(L.dyd_dedup_dev(h.data_ptr(), N, 0, keep.data_ptr(), sp), "k4")), kept=int(keep.sum().item()))
    stage("K3 hash128(ref) + K5 isin", int(rdata.numel()) + 8 * (R + 1) + 16 * R + 16 * N + N + 16 * R,
          timeit(lambda: (ck(L.dyd_hash128_dev(rdata.data_ptr(), roff.data_ptr(), R, hr.data_ptr(), sp), "k3r"),
                          ck(L.dyd_isin_dev(h.data_ptr(), N, hr.data_ptr(), R, hit.data_ptr(), sp), "k5"))), hits=int(hit.sum().item()))
    del data, off, rdata, roff, h, hr, keep, hit, ids
    stage("K1+K2 fused (poly->bbox + IoU flag)", 16 * P + 4 * (B + 1) + 48 * B + 4 * (N + 1) + N, timeit(tab["fused"]))
    # K6: one record per box; catA = c0..c9, catB = c10..c17, c18/c19 unclassified (SURVEY §8d rules); every category is
    # shuffled with the same seed (reference :800)
    labels = tab["label"]
    cat = torch.where(labels < 10, 0, torch.where(labels < 18, 1, -1)).to(torch.int32).contiguous()
    sizes = [int((cat == c).sum().item()) for c in (0, 1)]
    n_train = torch.tensor([int(s * 0.8) for s in sizes], dtype=torch.int64, device=dev)
    n_val = torch.tensor([int(s * 0.1) for s in sizes], dtype=torch.int64, device=dev)
    split = torch.empty(B, dtype=torch.uint8, device=dev)
    pos = torch.empty(B, dtype=torch.int64, device=dev)
    if hasattr(L, "dyd_split_ids_seeded_dev") and os.environ.get("DYD_BENCH_HOST_PERM") != "1":
        h_sizes = np.asarray(sizes, np.int64)
        h_tr, h_va = n_train.cpu().numpy(), n_val.cpu().numpy()
        ms = timeit(lambda: ck(L.dyd_split_ids_seeded_dev(cat.data_ptr(), B, 42, h_sizes.ctypes.data, h_tr.ctypes.data, h_va.ctypes.data,
                                                          2, None, split.data_ptr(), pos.data_ptr(), sp), "k6 seeded"))
        stage("K8 permutations (MT19937 stream + rejection resolve + radix sort, on the device) + K6 split ids (records = boxes)", 21 * B, ms,
              records=B, permutation="device")
    else:
        th = time.perf_counter()
        perm_np = np.concatenate([_native.mt19937_permutation(42, s) for s in sizes])
        host_perm_ms = (time.perf_counter() - th) * 1e3
        perm = torch.from_numpy(perm_np).to(dev)
        del perm_np
        cat_off = torch.tensor([0, sizes[0], sizes[0] + sizes[1]], dtype=torch.int64, device=dev)
        ms = timeit(lambda: ck(L.dyd_split_ids_dev(cat.data_ptr(), B, perm.data_ptr(), cat_off.data_ptr(), n_train.data_ptr(),
                                                   n_val.data_ptr(), 2, split.data_ptr(), pos.data_ptr(), sp), "k6"))
        stage("host MT19937 permutations (sequential Fisher-Yates)", 8 * B, host_perm_ms, records=B, permutation="host")
        stage("K6 split ids (records = boxes)", 21 * B, ms, records=B)
    total = sum(s["ms"] for s in stages)
    return {"config": "configs[2]: 10M rows full pipeline (dedup + ref-filter + poly->bbox + IoU + split) on 1 MI355X, every stage on "
                      "the full resident table, permutation included",
            "rows": N, "total_ms": round(total, 3), "rows_per_s": N / total * 1e3, "stages": stages}


def sharded_dedup(rows, rank, world, dev, reps=3, scaling="weak"):
    """configs[3]'s exchange on this rank's `rows` rows of a world*rows table: K3 hash of the URL column, K4 on the shard, ONE
    all-gather of the locally unique 16-B keys (RCCL over xGMI), K4 on this rank's hash slice of the gathered keys and one all-reduce
    of the verdict bytes; then the reference filter (reference keys sharded, gathered once, K5).  Wall-clock per stage with a
    device synchronisation on both sides, the collectives on their own; median of `reps`; every rank's numbers are reported."""
    import torch
    import torch.distributed as dist
    from deal_yolo_daya_amd import distributed as D

    ops = D.HipOps(dev)
    N = rows * world
    g = torch.Generator(device=dev).manual_seed(900 + rank)
    ids = torch.randint(0, int(0.9 * N) + 1, (rows,), generator=g, device=dev, dtype=torch.int64)     # ~40 % duplicates globally
    prefix, suffix = b"http://img.example/", b".jpg"
    width = len(prefix) + 10 + len(suffix)

    def url_bytes(v):
        n = v.numel()
        out = torch.empty((n, width), dtype=torch.uint8, device=dev)
        out[:, :len(prefix)] = torch.tensor(list(prefix), dtype=torch.uint8, device=dev)
        out[:, len(prefix) + 10:] = torch.tensor(list(suffix), dtype=torch.uint8, device=dev)
        v = v.clone()
        for k in range(9, -1, -1):
            out[:, len(prefix) + k] = (v % 10 + 48).to(torch.uint8)
            v //= 10
        return out.reshape(-1), torch.arange(n + 1, device=dev, dtype=torch.int64) * width

    data, off = url_bytes(ids)
    del ids
    ref_all = torch.arange(0, int(0.9 * N) + 1, 10, device=dev, dtype=torch.int64)
    rlo, rhi = D.shard_bounds(int(ref_all.numel()), world, rank)
    rdata, roff = url_bytes(ref_all[rlo:rhi])
    del ref_all

    def sync():
        torch.cuda.synchronize(dev)

    runs = []
    for _ in range(reps + 1):
        t = {}
        sync(); a = time.perf_counter()
        h = ops.hash128(data, off)
        sync(); t["k3_ms"] = (time.perf_counter() - a) * 1e3
        tm = {}
        a = time.perf_counter()
        keep = D.dedup_keys_sharded(h, "first", ops, timings=tm)
        sync(); t["dedup_total_ms"] = (time.perf_counter() - a) * 1e3
        t["dedup_collectives_ms"] = tm["collective_s"] * 1e3
        t["gathered_keys"], t["local_unique_keys"], t["slice_keys"] = tm["gathered_keys"], tm["local_unique_keys"], tm["slice_keys"]
        t["verdict_bytes"] = tm.get("verdict_bytes", 0)
        a = time.perf_counter()
        hr = ops.hash128(rdata, roff)
        sync(); b = time.perf_counter()
        all_ref, _ = D.all_gather_rows(hr)
        sync(); c = time.perf_counter()
        hit = ops.isin(h, all_ref)
        ops.check_status()
        sync(); d = time.perf_counter()
        t["ref_k3_ms"], t["ref_allgather_ms"], t["ref_k5_ms"] = (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3
        t["ref_table_keys"] = int(all_ref.shape[0])
        t["kept_local"], t["ref_hits_local"] = int(keep.sum().item()), int(hit.sum().item())
        runs.append(t)
    runs = runs[1:]
    med = {k: float(np.median([r[k] for r in runs])) for k in runs[0]}
    step_ms = med["k3_ms"] + med["dedup_total_ms"] + med["ref_k3_ms"] + med["ref_allgather_ms"] + med["ref_k5_ms"]
    mine = torch.tensor([step_ms, med["dedup_total_ms"], med["dedup_collectives_ms"], med["ref_k5_ms"], med["local_unique_keys"],
                         med["slice_keys"], med["ref_table_keys"], med["kept_local"], med["ref_hits_local"]], dtype=torch.float64, device=dev)
    every = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)
    if dist.get_backend() == "gloo":
        hc, ec = mine.cpu(), every.cpu()
        dist.all_gather_into_tensor(ec, hc); every = ec
    else:
        dist.all_gather_into_tensor(every, mine)
    every = every.cpu().numpy().reshape(world, -1)
    step_max = float(every[:, 0].max())
    return {"config": "configs[3] exchange: sharded URL dedup + reference filter, rows sharded contiguously, one all-gather of the "
                      "locally unique keys + one all-reduce of verdict bytes (dedup) and one all-gather of the reference keys",
            "scaling": scaling, "rows_per_gpu": rows, "rows_total": N, "world": world, "backend": dist.get_backend(),
            "stages_ms_rank0": {k: round(v, 3) for k, v in med.items() if k.endswith("_ms")},
            "allgather_bytes_dedup": int(med["gathered_keys"]) * 16, "allreduce_bytes_dedup": int(med["verdict_bytes"]),
            "per_rank": [{"rank": r, "step_ms": round(float(e[0]), 3), "dedup_total_ms": round(float(e[1]), 3), "dedup_collectives_ms": round(float(e[2]), 3),
                          "ref_k5_ms": round(float(e[3]), 3), "local_unique_keys": int(e[4]), "dedup_slice_keys": int(e[5]),
                          "ref_table_keys": int(e[6])} for r, e in enumerate(every)],
            "kept_rows_total": int(every[:, 7].sum()), "ref_hits_total": int(every[:, 8].sum()),
            "step_ms_max_over_ranks": round(step_max, 3), "rows_per_s": N / step_max * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ramp-ms", type=float, default=400.0,
                    help="untimed launches before the W warm-up steps until this much GPU time has passed: the card idles at "
                         "a low shader clock while the inputs are generated, and a handful of launches do not bring it up")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (overrides the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="largest CPU-baseline sample (0 = skip); 10000 rows are timed as well")
    ap.add_argument("--host-rows", type=int, default=1_000_000, help="rows of the host-inclusive DataFrame run (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=1, help="1 = also time configs[2]'s full pipeline per stage (c3, N=1)")
    ap.add_argument("--pipeline-cpu-rows", type=int, default=20000, help="rows of the CPU port's run of the five steps beside host_inclusive.pipeline (0 = skip)")
    ap.add_argument("--dense", type=int, default=1, help="1 = also time configs[4]'s table (1M rows x 256 boxes) with its own roofline object (c3, N=1)")
    ap.add_argument("--dense-steps", type=int, default=10)
    ap.add_argument("--path-rows", type=int, default=300000, help="rows of the CSV path-in / path-out legs (SURVEY §8d region 3; 0 = skip)")
    ap.add_argument("--exchange", type=int, default=1,
                    help="1 = also time configs[3]'s sharded dedup / reference filter with its collectives (after the K steps): weak (rows per GPU as "
                         "the workload says) at N > 1 or with --workload c4, strong (--strong-rows in total) at every N")
    ap.add_argument("--strong-rows", type=int, default=100_000_000, help="rows of the fixed table of the strong-scaling legs (exchange at every N, fused launch at N > 1; 0 = skip)")
    ap.add_argument("--strong-steps", type=int, default=20)
    ap.add_argument("--fused-variant", type=int, default=-1, help="A/B only: force a kernel variant of the fused launch (dyd_set_option)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    # rehearsal knobs (one-GPU box): DYD_BENCH_DEVICE pins every rank to one card, DYD_BENCH_BACKEND=gloo
    # avoids RCCL's one-rank-per-GPU rule; the driver's real runs leave both unset (RCCL, rank = GPU)
    dev_index = int(os.environ.get("DYD_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DYD_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from deal_yolo_daya_amd import _native, synth
    L = _native.load_library()
    _native.check(L.dyd_init(dev_index), "dyd_init")
    L = _native.lib()
    ck = _native.check

    if args.fused_variant >= 0:
        ck(L.dyd_set_option(b"fused_variant", args.fused_variant), "dyd_set_option")
    rows, bpr, desc = WORKLOADS[args.workload]
    if args.rows:
        rows = args.rows
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()

    def resident(rows, bpr, steps, warmup, ramp_ms, seed_base):
        """the synthetic batch for this rank, drawn on the device in chunks and resident in HBM; W warm-up steps, then EXACTLY
        `steps` timed steps of the fused launch, HIP events around each on the launch stream"""
        chunk = GEN_CHUNK if bpr is None else 100_000
        xy_p, npts_p, nbox_p, lab_p = [], [], [], []
        for ci, start in enumerate(range(0, rows, chunk)):
            d = synth.generate_device(min(chunk, rows - start), seed_base + ci, dev, boxes_per_row=bpr)
            xy_p.append(d["xy"])
            npts_p.append(torch.diff(d["pt_off"]))
            nbox_p.append(torch.diff(d["box_off"]))
            lab_p.append(d["label"])
            del d
        xy = torch.cat(xy_p); del xy_p
        npts = torch.cat(npts_p); nbox = torch.cat(nbox_p); label = torch.cat(lab_p)
        del npts_p, nbox_p, lab_p
        P, B, N = int(xy.shape[0]), int(npts.shape[0]), int(nbox.shape[0])
        if P >= 2 ** 31:
            raise SystemExit("points per GPU exceed int32 offsets; lower --rows")
        pt_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        pt_off[1:] = torch.cumsum(npts, 0, dtype=torch.int64).to(torch.int32)
        box_off = torch.zeros(N + 1, dtype=torch.int32, device=dev)
        box_off[1:] = torch.cumsum(nbox, 0, dtype=torch.int64).to(torch.int32)
        del npts, nbox
        out_box = torch.empty((B, 4), dtype=torch.float64, device=dev)
        out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
        out_high = torch.empty(N, dtype=torch.uint8, device=dev)
        torch.cuda.empty_cache()
        torch.cuda.synchronize()

        def fused():
            ck(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, P, MIN_BOXES, THR,
                                        out_box.data_ptr(), out_arg.data_ptr(), out_high.data_ptr(), sp), "dyd_bbox_iou_fused_dev")

        ramp_launches = 0
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < ramp_ms:      # clock ramp, outside the W + K steps
            for _ in range(4):
                fused()
            torch.cuda.synchronize()
            ramp_launches += 4
        for _ in range(warmup):
            fused()
        torch.cuda.synchronize()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):          # EXACTLY K timed steps
            ev[k][0].record(stream); fused(); ev[k][1].record(stream)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        k_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        return {"P": P, "B": B, "N": N, "elapsed": elapsed, "kernel_ms": k_ms, "ramp_launches": ramp_launches,
                "high_rows": int(out_high.sum().item()), "label": label, "fused": fused,
                "keep": (xy, pt_off, box_off, out_box, out_arg, out_high)}

    def roofline_of(res, workload, rows):
        """algorithmic bytes of one launch / its HIP-event time, and the PMC traffic of the same launch from the committed passes"""
        P, B, N = res["P"], res["B"], res["N"]
        alg_bytes = 16 * P + 4 * (B + 1) + 48 * B + 4 * (N + 1) + N     # K1's bytes + K2's offsets and flags; the boxes reach K2 through LDS
        achieved = alg_bytes / (res["kernel_ms"] * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: separate rocprofv3 --pmc passes over this same command (tools/gpu_profile.sh ->
        # tools/collect_profiles.py), FETCH_SIZE doubled as the microarch guide prescribes for gfx950.  It is NOT measured in
        # this run: the value is read from the committed summary of those passes and only for the workload they ran.
        traffic, traffic_src = None, None
        for tname in (("r03_c5_traffic.json", "r02_c5_traffic.json") if workload == "c5" else ("k12_traffic.json",)):
            tf = os.path.join(REPO, "profiles", tname)
            if not os.path.exists(tf):
                continue
            with open(tf) as fh:
                tj = json.load(fh)
            same = (tj.get("rows_per_gpu") == rows and tj.get("workload") == workload) or \
                   (workload == "c5" and tj.get("algorithmic_bytes_per_launch") == alg_bytes)
            if same:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{tname} <- {tj.get('source', 'rocprofv3 --pmc passes of this command')} (not measured in this run)"
                break
        return {"bound": "hbm", "kernel": ("k12_wave_kernel (fused K1+K2)" if B <= 32 * N else
                                           "k12_wave_dense_kernel (fused K1+K2, rows of 40..256 boxes sorted and swept)"),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                "frac_of_measured_achievable_6290": achieved / 6290.0}

    main_res = resident(rows, bpr, args.steps, args.warmup, args.ramp_ms, synth.SEED + 1000 * rank)
    elapsed = main_res["elapsed"]
    P, B, N = main_res["P"], main_res["B"], main_res["N"]
    label, fused = main_res["label"], main_res["fused"]
    main_info = {k: main_res[k] for k in ("P", "B", "N", "kernel_ms", "ramp_launches")}   # what the line needs once the tensors are gone
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist.get_backend() == "gloo":
            tc = tt.cpu(); dist.all_reduce(tc, op=dist.ReduceOp.MAX); tt = tc
        else:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    high_rows = main_res["high_rows"]
    exchange, exchange_strong = None, None
    if args.exchange:                                                # every rank takes part; rank 0 reports
        if world == 1 and not dist.is_initialized():                  # one GPU: a group of one, over RCCL like the real thing
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            backend = os.environ.get("DYD_BENCH_BACKEND", "nccl")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
        if world > 1 or args.workload == "c4":
            exchange = sharded_dedup(rows, rank, world, dev, scaling="weak")
        if args.strong_rows > 0:                                      # configs[3] as a FIXED table of --strong-rows rows cut into `world` shards
            exchange_strong = sharded_dedup(args.strong_rows // world, rank, world, dev, reps=2, scaling="strong")

    fused_strong = None
    if world > 1 and args.strong_rows > 0 and args.workload == "c3":
        # configs[3] as a FIXED table: --strong-rows rows (100 M) cut into `world` contiguous shards, the same fused launch on each rank's
        # shard (12.5 M rows per GPU at N = 8: configs[3] itself); max over ranks; together with sharded_exchange_strong the strong-scaling
        # series of the whole configuration.  Skipped where a shard would not fit: 2.9 KB per row resident, and about twice that while
        # the generator's chunks are concatenated (so 100 M rows run from N = 4 on; at N = 2 a shard is 145 GB before the doubling).
        srows = args.strong_rows // world
        fits = srows * 2900 * 2.2 < 250e9
        if fits:
            del label, fused
            main_res.clear()
            torch.cuda.empty_cache()
            # every rank must take the same branch (the leg ends in a collective): the shard has to fit on ALL of them, as measured now
            ok = torch.tensor([1 if torch.cuda.mem_get_info(dev)[0] > srows * 2900 * 2.2 else 0], dtype=torch.int32,
                              device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            fits = bool(int(ok.item()))
            if not fits:
                label = fused = None
        if fits:
            sres = resident(srows, None, args.strong_steps, 3, 200.0, synth.SEED + 7000 + 1000 * rank)
            st = torch.tensor([sres["elapsed"]], dtype=torch.float64, device=dev)
            if dist.get_backend() == "gloo":
                tc = st.cpu(); dist.all_reduce(tc, op=dist.ReduceOp.MAX); st = tc
            else:
                dist.all_reduce(st, op=dist.ReduceOp.MAX)
            s_elapsed = float(st.item())
            fused_strong = {"config": "configs[3] as a fixed table cut into N shards: the fused K1+K2 launch on every rank's shard, max over ranks",
                            "scaling": "strong", "rows_total": srows * world, "rows_per_gpu": srows, "steps": args.strong_steps,
                            "ms_per_step": s_elapsed * 1e3 / args.strong_steps, "kernel_ms_rank0": sres["kernel_ms"],
                            "rows_per_s": srows * world * args.strong_steps / s_elapsed,
                            "roofline_rank0": roofline_of(sres, "strong", srows)}
            sres.clear()
            label = fused = None
    if rank == 0:
        ms_step = elapsed * 1e3 / args.steps
        k_ms = main_info["kernel_ms"]
        line = {
            "metric": "annotation rows/sec through poly->bbox + IoU-filter path",
            "value": rows * world * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (drawn on the device: synth.generate_device, the distributions of SURVEY §8d)",
            "config": {"workload": desc, "rows_per_gpu": rows, "boxes_per_gpu": B, "points_per_gpu": P,
                       "min_boxes": MIN_BOXES, "iou_threshold": THR, "high_rows_rank0": high_rows,
                       "launch": "fused K1+K2 (dyd_bbox_iou_fused_dev)" + (f", forced variant {args.fused_variant}" if args.fused_variant >= 0 else ""), "kernel_ms": k_ms,
                       "clock_ramp_launches_before_warmup": main_info["ramp_launches"], "device": _native.device_name()},
            "roofline": roofline_of(main_info, args.workload, rows),
        }
        line["cpu_baseline"] = None
        line["host_inclusive"] = None
        line["full_pipeline"] = None
        line["dense"] = None
        line["sharded_exchange"] = exchange
        line["sharded_exchange_strong"] = exchange_strong
        line["fused_strong"] = fused_strong
        if world == 1:
            if args.pipeline and args.workload == "c3":
                line["full_pipeline"] = full_pipeline({"N": N, "B": B, "P": P, "label": label, "fused": fused}, dev, L, ck, sp)
            del label, fused
            main_res.clear()
            torch.cuda.empty_cache()
            if args.dense and args.workload == "c3":
                # configs[4] scaled to one launch: 1 M rows x 256 boxes (44 GB), the same fused entry (its DENSE instantiation), timed
                # like the headline: HIP events around every launch, its own roofline object
                drows, dbpr, ddesc = WORKLOADS["c5"]
                dres = resident(drows, dbpr, args.dense_steps, 2, 200.0, synth.SEED + 1000 * rank)   # the table `--workload c5` draws
                line["dense"] = {"workload": ddesc, "rows": drows, "boxes": dres["B"], "points": dres["P"], "steps": args.dense_steps,
                                 "ms_per_step": dres["elapsed"] * 1e3 / args.dense_steps, "kernel_ms": dres["kernel_ms"],
                                 "rows_per_s": drows * args.dense_steps / dres["elapsed"], "boxes_per_s": dres["B"] * args.dense_steps / dres["elapsed"],
                                 "high_rows": dres["high_rows"], "roofline": roofline_of(dres, "c5", drows)}
                dres.clear()
                torch.cuda.empty_cache()
            if args.cpu_sample > 0:
                sizes = sorted({min(10000, args.cpu_sample), args.cpu_sample})
                line["cpu_baseline"] = cpu_baseline(sizes)
            if args.host_rows > 0:
                df, ref, gen_s = host_table(args.host_rows, dev)
                hi = host_inclusive(df, gen_s)
                if line["cpu_baseline"]:
                    hi["vs_cpu_baseline"] = hi["value"] / line["cpu_baseline"]["value"]
                if args.pipeline:
                    hi["pipeline"] = host_pipeline(df, ref, min(args.pipeline_cpu_rows, args.host_rows))
                if args.path_rows > 0:
                    hi["path_io"] = path_io(df, min(args.path_rows, args.host_rows))
                line["host_inclusive"] = hi
        print(json.dumps(line, ensure_ascii=False))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
