#!/usr/bin/env python3
"""bench.py — annotation rows/s through the poly->bbox + IoU-filter path on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (K1 polygon->bbox, then K2 all-pairs IoU flag on K1's
boxes; by default both run inside ONE fused launch) over one batch of synthetic rows that is
already resident in HBM.  Default workload =
BASELINE.json configs[1]: 1M rows per GPU, <=32 boxes/image (weak scaling: every rank owns its
own 1M-row shard, no data-path collective — rows are independent).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     : dominant kernel (the fused K1+K2 kernel; K1 with --fused 0) — algorithmic bytes /
                 HIP-event time vs 8 TB/s HBM peak
  cpu_baseline : the CPU port of the reference path (oracle/steps.py) timed on this box's host,
                 rank 0 at N=1 only, on a bounded sample of the same synthetic rows.
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
    "c3": (10_000_000, None, "configs[2] kernels K1+K2 only: 10M rows (<=32 boxes/img) per GPU"),
    "c5": (1_000_000, 256, "configs[4] scaled: dense-box stress, 256 boxes/img, 4-pt polygons"),
}


def cpu_baseline(sample_rows: int):
    """Time the reference algorithm's CPU port on `sample_rows` synthetic rows (1 core)."""
    import pandas as pd  # noqa: F401
    from deal_yolo_daya_amd import synth
    from oracle import steps as osteps

    t = synth.generate(sample_rows, seed=synth.SEED)
    df = synth.to_frame(t)
    t0 = time.perf_counter()
    kept, projected, _ = osteps.replace_frame(df)
    t1 = time.perf_counter()
    high, other = osteps.iou_filter_frame(projected, MIN_BOXES, THR)
    t2 = time.perf_counter()
    total = t2 - t0
    return {
        "value": sample_rows / total,
        "unit": "rows/s",
        "cores": 1,
        "kind": "port",
        "sample": (f"{sample_rows} synthetic rows (same generator/seed as the GPU batch), in-memory "
                   f"DataFrame: replace step {t1 - t0:.2f}s + IoU step {t2 - t1:.2f}s, json.loads/dumps "
                   f"included, CSV I/O excluded; host has {os.cpu_count()} logical cores, 1 used"),
        "high_rows": int(len(high)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed launches before the W warm-up steps until this much GPU time has passed: the card idles at "
                         "a low shader clock while the inputs are generated on the host, and 3 launches (2 ms) do not bring it up")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (overrides the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="rows for the CPU baseline (0 = skip)")
    ap.add_argument("--fused", type=int, default=1,
                    help="1 = one fused K1+K2 launch (dyd_bbox_iou_fused_dev, default); 0 = K1 launch then K2 launch")
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

    rows, bpr, desc = WORKLOADS[args.workload]
    if args.rows:
        rows = args.rows
    # ---- synthetic batch for this rank, generated in chunks on the host, resident in HBM --------
    chunk = 1_000_000 if bpr is None else 100_000
    xy_parts, npts_parts, nbox_parts = [], [], []
    for ci, start in enumerate(range(0, rows, chunk)):
        n = min(chunk, rows - start)
        if bpr is None:
            t = synth.generate(n, seed=synth.SEED + 1000 * rank + ci)
        else:
            t = synth.generate(n, seed=synth.SEED + 1000 * rank + ci, boxes_per_row=bpr)
        xy_parts.append(torch.from_numpy(t.xy).to(dev))
        npts_parts.append(torch.from_numpy(np.diff(t.pt_off).astype(np.int32)).to(dev))
        nbox_parts.append(torch.from_numpy(np.diff(t.box_off).astype(np.int32)).to(dev))
        del t
    xy = torch.cat(xy_parts)
    del xy_parts
    npts = torch.cat(npts_parts)
    nbox = torch.cat(nbox_parts)
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
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def k1():
        _native.check(L.dyd_bbox_minmax_dev(xy.data_ptr(), pt_off.data_ptr(), B, P, out_box.data_ptr(),
                                            out_arg.data_ptr(), sp), "dyd_bbox_minmax_dev")

    def k2():
        _native.check(L.dyd_iou_any_ge_dev(out_box.data_ptr(), box_off.data_ptr(), N, B, MIN_BOXES, THR,
                                           out_high.data_ptr(), None, sp), "dyd_iou_any_ge_dev")

    def fused():
        _native.check(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, P,
                                               MIN_BOXES, THR, out_box.data_ptr(), out_arg.data_ptr(),
                                               out_high.data_ptr(), sp), "dyd_bbox_iou_fused_dev")

    def barrier():
        if world > 1:
            dist.barrier()

    ramp_launches = 0
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:      # clock ramp, outside the W + K steps
        for _ in range(16):
            fused() if args.fused else (k1(), k2())
        torch.cuda.synchronize()
        ramp_launches += 16
    for _ in range(args.warmup):
        if args.fused:
            fused()
        else:
            k1(); k2()
    torch.cuda.synchronize()

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):          # EXACTLY K timed steps
        if args.fused:
            ev[s][0].record(stream); fused(); ev[s][2].record(stream)
        else:
            ev[s][0].record(stream); k1(); ev[s][1].record(stream); k2(); ev[s][2].record(stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    high_rows = int(out_high.sum().item())

    if rank == 0:
        ms_step = elapsed * 1e3 / args.steps
        if args.fused:
            k1_ms = float(np.mean([e[0].elapsed_time(e[2]) for e in ev]))
            k2_ms = 0.0
        else:
            k1_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
            k2_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes over this same
        # command (tools/gpu_profile.sh -> tools/collect_profiles.py), FETCH_SIZE doubled as the
        # microarch guide prescribes for gfx950; reported only for the workload it was measured on.
        traffic = None
        tf = os.path.join(REPO, "profiles", "k1_traffic.json")
        if os.path.exists(tf):
            with open(tf) as fh:
                tj = json.load(fh)
            if (tj.get("rows_per_gpu") == rows and tj.get("workload") == args.workload
                    and bool(tj.get("fused")) == bool(args.fused)):
                traffic = tj["traffic_bytes_per_launch"]
        k1_bytes = 16 * P + 4 * (B + 1) + 48 * B
        k2_bytes = 32 * B + 4 * (N + 1) + N
        # fused launch: K2's 32*B box read is not compulsory traffic (the boxes were just produced)
        alg_bytes = (k1_bytes + 4 * (N + 1) + N) if args.fused else k1_bytes
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
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
            "data": "synthetic",
            "config": {"workload": desc, "rows_per_gpu": rows, "boxes_per_gpu": B, "points_per_gpu": P,
                       "min_boxes": MIN_BOXES, "iou_threshold": THR, "high_rows_rank0": high_rows,
                       "launch": "fused K1+K2" if args.fused else "K1 then K2",
                       "k1_ms": k1_ms, "k2_ms": k2_ms, "clock_ramp_launches_before_warmup": ramp_launches,
                       "k2_gbs": (k2_bytes / (k2_ms * 1e-3) / 1e9) if k2_ms else None,
                       "device": _native.device_name()},
            "roofline": {"bound": "hbm", "kernel": "k1_bbox_lds" if not args.fused else "k12_wave_kernel (fused K1+K2)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "frac_of_measured_achievable_6290": achieved / 6290.0},
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line, ensure_ascii=False))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
