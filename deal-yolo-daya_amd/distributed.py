"""Multi-GPU (one process per GPU) execution of the hot path — SURVEY §8e.

Rows shard contiguously: rank r owns global rows [r*N/G, (r+1)*N/G), so "first occurrence" is the
lowest (rank, local index).  The poly->bbox (K1) and IoU (K2) stages are independent per row and
need no communication.  The two set-valued stages exchange hashes exactly once:

    dedup       K3 hash local rows -> K4 on the shard alone (local keep-mask) -> ONE all-gather of the shard's
                locally-UNIQUE 16-B keys (RCCL over xGMI; an 8-byte count per rank goes ahead of it) -> every rank
                settles the 1/G slice of the gathered keys whose hash falls to it with K4 (the gathered array is in
                (rank, row) order, so "first occurrence" inside it is the lowest (rank, local index) and no row index
                travels) -> the verdicts, one byte per gathered key, are merged with one all-reduce.  A rank inserts
                ~U/G keys, whatever its place in the row order
    ref filter  K3 hash local main rows and local reference rows -> ONE all-gather of the
                reference keys -> K5 probe of the local main keys
    split       ONE all-gather of per-rank per-category counts (n_cat x 8 B) -> each rank's
                in-category ranks start at the sum of the lower ranks' counts; the permutation of each GLOBAL
                category size is made on every rank's device from the seed (K8, dyd_split_ids_seeded_dev): nothing
                but the counts crosses ranks.

    replace+IoU rows are independent: every rank runs the fused pass on its share of the rows, the shares balanced by the
                annotation cells' bytes (shard_bounds_by_weight); ONE all-gather of five counts per rank for the totals

    label lines rows are independent (K7 per shard); ONE all-gather of the shards' text sizes (8 B each) turns the
                local byte offsets into offsets inside the concatenated text of all ranks

``torch.distributed`` is the plumbing (backend "nccl" is RCCL on ROCm; tests use "gloo" on CPU).
The device work goes through an ``ops`` object: ``HipOps`` (below) drives the ``_dev`` entry
points of libdyd_gfx950.so on tensors resident in HBM; the CPU test-suite injects its own.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import flatten as _fl

_KEEP = {"first": 0, "last": 1, False: 2}


def shard_bounds(n: int, world: int, rank: int) -> tuple:
    """Contiguous row range of `rank`: sizes differ by at most one row."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def shard_bounds_by_weight(weights, world: int, rank: int) -> tuple:
    """Contiguous row range of `rank` holding about 1/world of sum(weights) ("shards are balanced by box count, not
    row count, when n_i is skewed").  The cut after row i falls where the running sum first reaches k/world of the total, so the
    ranges tile [0, n) in row order — "first occurrence" stays the lowest (rank, local index).  All-zero weights: by rows."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if n == 0 or world <= 1:
        return (0, n)
    c = np.cumsum(w)
    if not c[-1] > 0:
        return shard_bounds(n, world, rank)
    cuts = np.searchsorted(c, c[-1] * np.arange(1, world) / world, side="left") + 1
    edges = np.concatenate([[0], np.minimum(cuts, n), [n]])
    edges = np.maximum.accumulate(edges)
    return int(edges[rank]), int(edges[rank + 1])


def annotation_weights(col) -> np.ndarray:
    """per row the bytes of its annotation cell (0 for a missing one) plus a constant for the row itself: what a row costs the
    replace -> IoU pass is proportional to its text (points and boxes are spelled out in it), and it is known before any parsing"""
    from . import pycells
    values = col.to_numpy() if hasattr(col, "to_numpy") else np.asarray(col, dtype=object)
    if pycells.available() and values.dtype == object:
        try:
            v = pycells.CellViews(values)
            return np.where(v.missing != 0, 0, v.len).astype(np.int64) + 64
        except UnicodeEncodeError:
            pass
    return np.fromiter((len(c) if isinstance(c, str) else 0 for c in values), dtype=np.int64, count=len(values)) + 64


def replace_and_filter_sharded(df, min_boxes: int = 2, iou_threshold: float = 0.98, backend=None, group=None) -> dict:
    """SURVEY §8e for the replace and IoU steps (a3, a4): rows are independent, so every rank runs the fused replace -> IoU pass
    (processor.replace_and_filter_frame: scan, ONE fused K1+K2 launch per part, emit) on its contiguous share of the table and
    nothing of the data path crosses ranks.  The shares are balanced by the annotation cells' bytes (`shard_bounds_by_weight` over
    `annotation_weights`), not by rows: a table whose dense images cluster would otherwise leave most ranks waiting for one.
    ONE all-gather of five counts per rank gives the totals.

    `df` is the whole table on every rank (each rank touches only its rows' cells).  -> {"bounds": (lo, hi), "frames": (kept,
    excluded, high, other) of the local rows with their original row labels, "totals": {rows, kept, excluded, high, other} over
    all ranks, "per_rank": [[rows, kept, excluded, high, other], ...]}"""
    from .core import processor as P

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds_by_weight(annotation_weights(df[P.ANNOTATION_COL]), world, rank)
    frames = P.replace_and_filter_frame(df.iloc[lo:hi], min_boxes, iou_threshold, backend)
    mine = [hi - lo] + [len(f) for f in frames]
    per_rank = [mine]
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.tensor(mine, dtype=torch.int64, device=dev)
        every = torch.empty(world * len(mine), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(every, t, group=group)
        per_rank = every.reshape(world, len(mine)).cpu().tolist()
    names = ("rows", "kept", "excluded", "high", "other")
    return {"bounds": (lo, hi), "frames": frames, "per_rank": per_rank,
            "totals": {k: int(sum(r[i] for r in per_rank)) for i, k in enumerate(names)}}


class HipOps:
    """Device stage on the rank's GPU through the C ABI (_dev entry points, torch tensors in HBM)."""

    def __init__(self, device=None):
        from . import _native

        self.native = _native
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.L = _native.load_library()
        _native.check(self.L.dyd_init(self.device.index or 0), "dyd_init")
        self.L = _native.lib()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def tensor(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def hash128(self, data: torch.Tensor, off: torch.Tensor) -> torch.Tensor:
        n = off.numel() - 1
        out = torch.empty((n, 2), dtype=torch.int64, device=self.device)
        if n:
            self.native.check(self.L.dyd_hash128_dev(data.data_ptr(), off.data_ptr(), n, out.data_ptr(), self._stream()),
                              "dyd_hash128_dev")
        return out

    def check_status(self):
        """device-side failures of the launches queued so far (hash table full ...): raises NativeError"""
        self.native.check(self.L.dyd_device_status(self._stream()), "dyd_device_status")

    def dedup_local(self, h: torch.Tensor, keep) -> torch.Tensor:
        """K4 over this rank's keys alone -> uint8 keep-mask"""
        out = torch.empty(h.shape[0], dtype=torch.uint8, device=self.device)
        if h.shape[0]:
            self.native.check(self.L.dyd_dedup_dev(h.data_ptr(), h.shape[0], _KEEP[keep], out.data_ptr(), self._stream()),
                              "dyd_dedup_dev")
        return out

    def split_ids_seeded(self, cat, seed, sizes, n_train, n_val, rank_base):
        """K8 + K6 for a shard: host arrays for the categories' sizes / cuts / rank bases, `cat` on the device"""
        n = cat.numel()
        split = torch.empty(n, dtype=torch.uint8, device=self.device)
        pos = torch.empty(n, dtype=torch.int64, device=self.device)
        if n:
            sizes, n_train, n_val, rank_base = (np.ascontiguousarray(a, dtype=np.int64) for a in (sizes, n_train, n_val, rank_base))
            self.native.check(self.L.dyd_split_ids_seeded_dev(cat.data_ptr(), n, int(seed), sizes.ctypes.data, n_train.ctypes.data,
                                                              n_val.ctypes.data, len(sizes), rank_base.ctypes.data, split.data_ptr(),
                                                              pos.data_ptr(), self._stream()), "dyd_split_ids_seeded_dev")
        return split, pos

    def dedup_global(self, all_h: torch.Tensor, first: int, n_local: int, keep) -> torch.Tensor:
        out = torch.empty(n_local, dtype=torch.uint8, device=self.device)
        if n_local:
            self.native.check(self.L.dyd_dedup_global_dev(all_h.data_ptr(), all_h.shape[0], first, n_local, _KEEP[keep],
                                                          out.data_ptr(), self._stream()), "dyd_dedup_global_dev")
        return out

    def isin(self, h: torch.Tensor, ref_h: torch.Tensor) -> torch.Tensor:
        out = torch.empty(h.shape[0], dtype=torch.uint8, device=self.device)
        if h.shape[0]:
            self.native.check(self.L.dyd_isin_dev(h.data_ptr(), h.shape[0], ref_h.data_ptr() if ref_h.shape[0] else None,
                                                  ref_h.shape[0], out.data_ptr(), self._stream()), "dyd_isin_dev")
        return out

    def split_ids_sharded(self, cat, perm, cat_off, n_train, n_val, rank_base):
        n = cat.numel()
        split = torch.empty(n, dtype=torch.uint8, device=self.device)
        pos = torch.empty(n, dtype=torch.int64, device=self.device)
        if n:
            self.native.check(self.L.dyd_split_ids_sharded_dev(
                cat.data_ptr(), n, perm.data_ptr(), cat_off.data_ptr(), n_train.data_ptr(), n_val.data_ptr(),
                n_train.numel(), rank_base.data_ptr(), split.data_ptr(), pos.data_ptr(), self._stream()),
                "dyd_split_ids_sharded_dev")
        return split, pos

    def permutation(self, seed: int, n: int) -> np.ndarray:
        return self.native.mt19937_permutation(seed, n)

    def yolo_lines(self, box4, row_off, sel, width, height, class_id):
        """K7 on tensors in HBM -> (text_off int64 [n+1], flag u8 [n], text u8 [total]) on the device"""
        import ctypes as C
        n = row_off.numel() - 1
        toff = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        flag = torch.zeros(max(n, 1), dtype=torch.uint8, device=self.device)[:n]
        total = C.c_int64()
        args = (box4.data_ptr(), row_off.data_ptr(), sel.data_ptr() if sel is not None else None, width.data_ptr(),
                height.data_ptr(), class_id.data_ptr(), n, int(box4.numel()) // 4, toff.data_ptr(), flag.data_ptr())   # box4 may come flat
        self.native.check(self.L.dyd_yolo_lines_dev(*args, None, 0, C.byref(total), self._stream()), "dyd_yolo_lines_dev")
        text = torch.empty(max(total.value, 1), dtype=torch.uint8, device=self.device)
        if total.value:
            self.native.check(self.L.dyd_yolo_lines_dev(*args, text.data_ptr(), total.value, C.byref(total), self._stream()),
                              "dyd_yolo_lines_dev")
        return toff, flag, text[:total.value]


def all_gather_rows(t: torch.Tensor, group=None) -> tuple:
    """Concatenate every rank's [n_r, ...] tensor in rank order with ONE data collective
    (all_gather_into_tensor on shards padded to the largest; shards differ by <= 1 row).
    -> (gathered [sum n_r, ...], counts per rank)"""
    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":     # one-GPU rehearsal of several ranks: gloo moves host memory
        out, counts = all_gather_rows(t.cpu(), group)
        return out.to(t.device), counts
    n_local = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts_t = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts_t, n_local, group=group)
    counts = counts_t.tolist()
    m = max(counts)
    padded = t
    if t.shape[0] != m:
        padded = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        padded[: t.shape[0]] = t
    out = torch.empty((world * m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == m for c in counts):
        return out, counts
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)]), counts


def _local_keys(col, ops, for_isin: bool = False, drop_na: bool = False) -> torch.Tensor:
    """K3 over this rank's cells -> [n, 2] int64 keys resident on the device."""
    if for_isin:
        data, off = _fl.column_str_bytes(col, drop_na=drop_na)
        na = None
    else:
        data, off, na = _fl.column_key_bytes(col)
    h = ops.hash128(ops.tensor(np.ascontiguousarray(data)), ops.tensor(off))
    if na is not None and na.any():
        h[ops.tensor(na)] = ops.tensor(_fl.NA_KEY.view(np.int64))
    return h


def dedup_keys_sharded(h: torch.Tensor, keep, ops, group=None, timings: dict | None = None) -> torch.Tensor:
    """Global keep-mask (bool tensor on h's device) of this rank's keys h [n, 2]: local K4, ONE all-gather of the locally unique
    keys, then every rank settles a 1/G SLICE of the gathered keys (by hash) with K4 and the verdicts travel back as one byte per
    gathered key (an all-reduce, a sixteenth of the all-gather's bytes).

    Round 2 let rank r probe its survivors against the keys of the ranks below it: rank G-1 built a table of 7/8 of all unique
    keys while rank 0 built none, so the stage took as long as its last rank and did not shrink with G.  The gathered array is in
    (rank, row) order, so "first occurrence" inside any subset of it IS the lowest (rank, local index): K4 with the same ``keep``
    on the keys of one hash slice, in gathered order, says for every key of the slice whether it survives globally.  Each rank
    therefore inserts ~U/G keys instead of up to U(G-1)/G."""
    import time as _t

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    local_keep = ops.dedup_local(h, keep).bool()
    if world == 1:
        if hasattr(ops, "check_status"):
            ops.check_status()
        if timings is not None:
            timings.update({"collective_s": timings.get("collective_s", 0.0), "gathered_keys": 0, "local_unique_keys": int(local_keep.sum()),
                            "slice_keys": 0, "verdict_bytes": 0})
        return local_keep
    uniq_mask = local_keep if keep in ("first", "last") else ops.dedup_local(h, "first").bool()
    uniq = h[uniq_mask].contiguous()                      # one key per distinct local value, in row order
    if timings is not None and h.is_cuda:
        torch.cuda.synchronize(h.device)
    t0 = _t.perf_counter()
    gathered, counts = all_gather_rows(uniq, group)
    if timings is not None:
        if h.is_cuda:
            torch.cuda.synchronize(h.device)
        timings["collective_s"] = timings.get("collective_s", 0.0) + (_t.perf_counter() - t0)
        timings["gathered_keys"] = int(sum(counts))
        timings["local_unique_keys"] = int(uniq.shape[0])
    start = int(sum(counts[:rank]))
    end = start + counts[rank]
    # this rank's slice of the gathered keys (the hash's first word modulo the world size), settled with K4 in gathered order
    mine = torch.remainder(gathered[:, 0], world) == rank
    where = torch.nonzero(mine).reshape(-1)
    dropped = torch.zeros(gathered.shape[0], dtype=torch.uint8, device=gathered.device)
    if where.numel():
        kept = ops.dedup_local(gathered[where].contiguous(), keep).bool()
        dropped[where] = (~kept).to(torch.uint8)
    if timings is not None:
        if h.is_cuda:
            torch.cuda.synchronize(h.device)
        timings["slice_keys"] = int(where.numel())
    t1 = _t.perf_counter()
    if dropped.is_cuda and dist.get_backend(group) == "gloo":   # one-GPU rehearsal of several ranks: gloo moves host memory
        host = dropped.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        dropped = host.to(dropped.device)
    else:
        dist.all_reduce(dropped, op=dist.ReduceOp.MAX, group=group)   # the slices are disjoint: MAX just merges them
    if timings is not None:
        if h.is_cuda:
            torch.cuda.synchronize(h.device)
        timings["collective_s"] = timings.get("collective_s", 0.0) + (_t.perf_counter() - t1)
        timings["verdict_bytes"] = int(dropped.numel())
    out = local_keep.clone()
    own = dropped[start:end].bool()                       # per locally unique key, in row order
    if keep in ("first", "last"):
        out[uniq_mask] = ~own                             # the locally kept rows ARE the locally unique ones
    else:
        # keep=False: a locally single key survives only if no other rank holds it; its verdict sits at its first local occurrence
        first_of = torch.full((h.shape[0],), False, dtype=torch.bool, device=h.device)
        first_of[uniq_mask] = ~own
        out = local_keep & first_of
    if hasattr(ops, "check_status"):
        ops.check_status()
    return out


def dedup_keep_mask_sharded(local_col, keep="first", ops=None, group=None) -> np.ndarray:
    """keep-mask of this rank's shard of ``drop_duplicates(keep=keep)`` over the GLOBAL column
    (reference core/processor.py:140-144)."""
    if keep not in _KEEP:
        raise ValueError('keep must be either "first", "last" or False')
    ops = ops or HipOps()
    h = _local_keys(local_col, ops)
    return dedup_keys_sharded(h, keep, ops, group).cpu().numpy().astype(bool)


def ref_hit_mask_sharded(local_main_col, local_ref_col, ops=None, group=None) -> np.ndarray:
    """``main.astype(str).isin(set(ref.dropna().astype(str)))`` for this rank's main rows, with
    the reference column sharded too (reference core/processor.py:194-198)."""
    ops = ops or HipOps()
    hm = _local_keys(local_main_col, ops, for_isin=True)
    hr = _local_keys(local_ref_col, ops, for_isin=True, drop_na=True)
    all_ref, _ = all_gather_rows(hr, group)
    hit = ops.isin(hm, all_ref)
    if hasattr(ops, "check_status"):
        ops.check_status()
    return hit.cpu().numpy().astype(bool)


def split_ids_sharded(local_cat: np.ndarray, n_cat: int, train_ratio=0.8, val_ratio=0.1, test_ratio=0.1,
                      random_seed: int = 42, ops=None, group=None) -> tuple:
    """(split id u8, shuffled position i64) for this rank's shard of the expanded rows; category
    ids are GLOBAL ids in [0, n_cat) (-1 = unclassified) (reference core/processor.py:796-806)."""
    from .core.processor import split_cut_sizes

    ops = ops or HipOps()
    rank = dist.get_rank(group)
    local_cat = np.ascontiguousarray(local_cat, dtype=np.int32)
    counts = np.bincount(local_cat[local_cat >= 0], minlength=n_cat).astype(np.int64)
    all_counts, _ = all_gather_rows(ops.tensor(counts.reshape(1, -1)), group)       # [world, n_cat]
    all_counts = all_counts.cpu().numpy()
    sizes = all_counts.sum(axis=0)
    rank_base = all_counts[:rank].sum(axis=0).astype(np.int64)
    cat_off = np.zeros(n_cat + 1, np.int64)
    np.cumsum(sizes, out=cat_off[1:])
    cuts = [split_cut_sizes(int(s), train_ratio, val_ratio, test_ratio) for s in sizes]
    n_train = np.asarray([c[0] for c in cuts], np.int64)
    n_val = np.asarray([c[1] for c in cuts], np.int64)
    if hasattr(ops, "split_ids_seeded"):                  # K8: the permutations never exist on the host
        split, pos = ops.split_ids_seeded(ops.tensor(local_cat), random_seed, sizes, n_train, n_val, rank_base)
    else:
        perm = (np.concatenate([ops.permutation(random_seed, int(s)) for s in sizes])
                if n_cat else np.zeros(0, np.int64))
        split, pos = ops.split_ids_sharded(ops.tensor(local_cat), ops.tensor(perm), ops.tensor(cat_off),
                                           ops.tensor(n_train), ops.tensor(n_val), ops.tensor(rank_base))
    return split.cpu().numpy(), pos.cpu().numpy()


def yolo_lines_sharded(box4, row_off, sel, width, height, class_id, ops=None, group=None) -> tuple:
    """Label lines (reference core/processor.py:1046-1054) of this rank's shard of the rows.
    -> (global byte offset of every local row [n+1], flag [n], local text, total bytes of all ranks): the rows'
    texts of all ranks, concatenated in rank order, are what one process would have produced."""
    ops = ops or HipOps()
    rank = dist.get_rank(group)
    toff, flag, text = ops.yolo_lines(ops.tensor(np.ascontiguousarray(box4, np.float64).reshape(-1)),
                                      ops.tensor(np.ascontiguousarray(row_off, np.int32)),
                                      None if sel is None else ops.tensor(np.ascontiguousarray(sel, np.uint8)),
                                      ops.tensor(np.ascontiguousarray(width, np.float64)),
                                      ops.tensor(np.ascontiguousarray(height, np.float64)),
                                      ops.tensor(np.ascontiguousarray(class_id, np.int32)))
    sizes, _ = all_gather_rows(ops.tensor(np.asarray([[int(text.numel())]], np.int64)), group)
    sizes = sizes.cpu().numpy().reshape(-1)
    return toff.cpu().numpy() + int(sizes[:rank].sum()), flag.cpu().numpy(), text.cpu().numpy().tobytes(), int(sizes.sum())
