"""Worker for tests/test_distributed_cpu.py: one rank of a gloo group on CPU.  The CPU oracle
stands in for the device stage (``OracleOps``), so the test covers the sharding, the collectives
and the global-index bookkeeping of deal_yolo_daya_amd.distributed."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle import lib as olib  # noqa: E402

_KEEP = {"first": 0, "last": 1, False: 2}


class OracleOps:
    """CPU stand-in with the method set of distributed.HipOps (tensors live on the CPU)."""

    def tensor(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def hash128(self, data, off):
        return torch.from_numpy(olib.hash128(data.numpy(), off.numpy()).view(np.int64))

    def dedup_local(self, h, keep):
        return torch.from_numpy(olib.dedup(h.numpy().view(np.uint64), _KEEP[keep]))

    def dedup_global(self, all_h, first, n_local, keep):
        mask = olib.dedup(all_h.numpy().view(np.uint64), _KEEP[keep])
        return torch.from_numpy(mask[first:first + n_local].copy())

    def isin(self, h, ref_h):
        return torch.from_numpy(olib.isin(h.numpy().view(np.uint64), ref_h.numpy().view(np.uint64)))

    def split_ids_sharded(self, cat, perm, cat_off, n_train, n_val, rank_base):
        # emulate the rank base by prepending `rank_base[c]` dummy rows of every category
        cat, base = cat.numpy(), rank_base.numpy()
        pre = np.repeat(np.arange(len(base), dtype=np.int32), base)
        split, pos = olib.split_ids(np.concatenate([pre, cat]), perm.numpy(), cat_off.numpy(), n_train.numpy(),
                                    n_val.numpy())
        return torch.from_numpy(split[len(pre):].copy()), torch.from_numpy(pos[len(pre):].copy())

    def permutation(self, seed, n):
        return olib.mt19937_permutation(seed, n)

    def yolo_lines(self, box4, row_off, sel, width, height, class_id):
        off, flag, text = olib.yolo_lines(box4.numpy(), row_off.numpy(), None if sel is None else sel.numpy(), width.numpy(),
                                          height.numpy(), class_id.numpy())
        return torch.from_numpy(off), torch.from_numpy(flag), torch.from_numpy(np.frombuffer(text, np.uint8).copy())


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pandas as pd
    from deal_yolo_daya_amd import distributed as D

    ops = OracleOps()
    rng = np.random.default_rng(123)                       # same table on every rank
    n = 5003
    ids = rng.integers(0, 1800, size=n)
    src = pd.Series([None if k % 97 == 0 else f"http://img.example/{k}.jpg" for k in ids.tolist()], dtype=object)
    ref = pd.Series([f"http://img.example/{k}.jpg" for k in range(0, 1800, 10)] + [None, "nan"], dtype=object)
    cat = rng.integers(-1, 3, size=n).astype(np.int32)

    lo, hi = D.shard_bounds(n, world, rank)
    rlo, rhi = D.shard_bounds(len(ref), world, rank)
    res = {"lo": lo, "hi": hi}
    for keep in ("first", "last", False):
        res[f"dedup_{keep}"] = D.dedup_keep_mask_sharded(src.iloc[lo:hi], keep, ops).astype(int).tolist()
    res["ref_hit"] = D.ref_hit_mask_sharded(src.iloc[lo:hi], ref.iloc[rlo:rhi], ops).astype(int).tolist()
    split, pos = D.split_ids_sharded(cat[lo:hi], 3, 0.8, 0.1, 0.1, 42, ops)
    res["split"], res["pos"] = split.tolist(), pos.tolist()
    boxes = np.round(rng.random((n, 4)) * 500, 1)
    boxes[:, 2:] += boxes[:, :2] + 1
    boxes[::50, 2] = boxes[::50, 0]                            # zero-width boxes: no line
    goff, gflag, gtext, gtotal = D.yolo_lines_sharded(boxes[lo:hi], np.arange(hi - lo + 1, dtype=np.int32), None,
                                                      np.full(hi - lo, 640.0), np.full(hi - lo, 480.0),
                                                      (np.arange(lo, hi) % 13).astype(np.int32), ops)
    res["yolo_off"], res["yolo_flag"], res["yolo_text"], res["yolo_total"] = goff.tolist(), gflag.tolist(), gtext.decode(), gtotal
    # replace -> IoU, shares balanced by annotation bytes: a table whose dense images sit at the front
    from deal_yolo_daya_amd import synth
    from deal_yolo_daya_amd.core import processor as P
    from helpers import OracleBackend
    dense = synth.to_frame(synth.generate(60, seed=31, boxes_per_row=40))
    sparse = synth.to_frame(synth.generate(400, seed=32))
    table = pd.concat([dense, sparse], ignore_index=True)
    table.loc[[5, 200, 459], P.ANNOTATION_COL] = None
    table.index = pd.Index(np.arange(len(table)) * 2 + 1)
    sharded = D.replace_and_filter_sharded(table, 2, 0.98, backend=OracleBackend())
    weights = D.annotation_weights(table[P.ANNOTATION_COL])
    res["rf_bounds"], res["rf_totals"], res["rf_per_rank"] = list(sharded["bounds"]), sharded["totals"], sharded["per_rank"]
    res["rf_weight"] = int(weights[sharded["bounds"][0]:sharded["bounds"][1]].sum())
    res["rf_weight_total"] = int(weights.sum())
    res["rf_labels"] = [f.index.tolist() for f in sharded["frames"]]
    res["rf_bbox"] = sharded["frames"][0][P.BBOX_COL].tolist()
    g, counts = D.all_gather_rows(torch.arange(lo, hi).reshape(-1, 1))
    res["gathered_ok"] = bool(torch.equal(g.flatten(), torch.arange(n))) and counts == [
        D.shard_bounds(n, world, r)[1] - D.shard_bounds(n, world, r)[0] for r in range(world)]
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
