#!/usr/bin/env python3
"""dyd_split_ids_seeded_dev alone (K8 permutations on the device + K6 split ids) on configs[2]'s record table: E records drawn with
the label distribution of SURVEY §8d (two categories of ~82.5 M each at the default 165 M + 10 % unclassified), HIP-event timed.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/gpu_profile_k8.sh)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=165_004_682)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=0, help="1 = compare positions with numpy's RandomState.permutation (slow at full size)")
    ap.add_argument("--single", type=int, default=0, help="1 = one category only (half the records): per-kernel times without a second category's kernels beside them")
    args = ap.parse_args()
    import torch
    from deal_yolo_daya_amd import _native

    dev = torch.device("cuda", 0)
    L = _native.lib()
    ck = _native.check
    B = args.records
    g = torch.Generator(device=dev).manual_seed(3)
    labels = torch.randint(0, 20, (B,), generator=g, device=dev, dtype=torch.int32)
    cat = torch.where(labels < 10, 0, torch.where(labels < 18, 1, -1)).to(torch.int32).contiguous()
    del labels
    n_cat = 2
    if args.single:
        cat = torch.where(cat == 0, 0, -1).to(torch.int32).contiguous()
        n_cat = 1
    sizes = np.asarray([int((cat == c).sum().item()) for c in range(n_cat)], np.int64)
    n_train = (sizes * 0.8).astype(np.int64)
    n_val = (sizes * 0.1).astype(np.int64)
    split = torch.empty(B, dtype=torch.uint8, device=dev)
    pos = torch.empty(B, dtype=torch.int64, device=dev)
    sp = torch.cuda.current_stream().cuda_stream

    def run():
        ck(L.dyd_split_ids_seeded_dev(cat.data_ptr(), B, 42, sizes.ctypes.data, n_train.ctypes.data, n_val.ctypes.data, n_cat, None,
                                      split.data_ptr(), pos.data_ptr(), sp), "dyd_split_ids_seeded_dev")

    run()
    torch.cuda.synchronize()
    ts, wall = [], []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        a.record(); run(); b.record(); b.synchronize()
        wall.append((time.perf_counter() - w0) * 1e3)
        ts.append(a.elapsed_time(b))
    out = {"what": "dyd_split_ids_seeded_dev (K8 + K6)", "records": B, "category_sizes": sizes.tolist(), "ms": [round(t, 3) for t in ts],
           "median_ms": round(float(np.median(ts)), 3), "wall_ms_median": round(float(np.median(wall)), 3),
           "alg_GB": round(21 * B / 1e9, 3), "GBs": round(21 * B / np.median(ts) / 1e6, 1)}
    if args.check:
        p = pos.cpu().numpy()
        c = cat.cpu().numpy()
        for k in range(n_cat):
            inv = np.empty(int(sizes[k]), np.int64)
            inv[np.random.RandomState(42).permutation(int(sizes[k]))] = np.arange(int(sizes[k]))
            assert np.array_equal(p[c == k], inv), f"category {k}: positions differ from numpy"
        out["matches_numpy"] = True
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
