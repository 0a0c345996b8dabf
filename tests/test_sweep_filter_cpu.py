"""The sort-and-sweep filter of csrc/k2_sweep.h, restated in numpy: whatever the exact f64 test (the oracle's calculate_iou,
reference core/processor.py:328-339) calls a hit must lie inside the window the filter leaves open — for either box of the pair as
the one that is met first in x1 order — and pass its y test.  This pins the ARGUMENT (window derivation, outward f32 bounds, key
truncation, limit slack); the kernels themselves are compared with the oracle on the GPU (tests/test_gpu_sweep.py)."""
import numpy as np
import pytest

from oracle import lib as olib


def _f32_order(f):
    b = np.asarray(f, np.float32).view(np.uint32)
    return np.where(b & np.uint32(0x80000000), ~b, b | np.uint32(0x80000000)).astype(np.uint32)


def _push(v, sign):
    """k2_filter.h f32_below / f32_above: the nearest f32, clamped on the far side, pushed out by 2^-23 of itself and 2^-120"""
    with np.errstate(over="ignore", invalid="ignore"):
        f = np.asarray(v, np.float64).astype(np.float32)
        fmax = np.float32(3.402823466e+38)
        f = np.minimum(f, fmax) if sign < 0 else np.maximum(f, -fmax)
        fused = (f.astype(np.float64) + sign * np.abs(f.astype(np.float64)) * 2.0 ** -23).astype(np.float32)   # fmaf: exact, rounded once
        return (fused + np.float32(sign) * np.float32(2.0 ** -120)).astype(np.float32)


def _records(box, thr, ib=8):
    """per box: key (x1 bound | index), limit, y interval — k2s_prepare"""
    x1, y1 = np.minimum(box[:, 0], box[:, 2]), np.minimum(box[:, 1], box[:, 3])
    x2, y2 = np.maximum(box[:, 0], box[:, 2]), np.maximum(box[:, 1], box[:, 3])
    tl = thr * 0.999
    lim = x2 - tl * (x2 - x1)
    im = np.uint32((1 << ib) - 1)
    key = (_f32_order(_push(x1, -1)) & ~im) | np.arange(len(box), dtype=np.uint32)
    limit = (_f32_order(_push(lim, +1)) + (im + np.uint32(1))) | im
    return key, limit, _push(y1, -1), _push(y2, +1)


def _hits(box, thr):
    n = len(box)
    out = []
    for i in range(n):
        for j in range(i + 1, n):
            pair = np.stack([box[i], box[j]])
            if olib.iou_any_ge(pair, np.array([0, 2], np.int32), 2, thr)[0]:
                out.append((i, j))
    return out


def _tables():
    rng = np.random.default_rng(1)
    t = {}
    c = rng.random((60, 2)) * [1920, 1080]
    wh = rng.random((60, 2)) * 90 + 1
    base = np.concatenate([c, c + wh], axis=1)
    for k in range(0, 60, 3):                                   # near copies at and around the threshold
        base[k + 1] = base[k]
        base[k + 1, 0] += (base[k, 2] - base[k, 0]) * [0.0, 0.0199, 0.0201, 0.005, 0.5][(k // 3) % 5]
        base[k + 2] = base[k][[2, 3, 0, 1]]                     # un-normalised corners
    t["uniform"] = base
    t["integers"] = np.round(base)
    t["beyond_f32"] = base * 4096.0 + 16777216.123
    t["negative"] = base - [2000, 1000, 2000, 1000]
    t["tiny"] = base * 1e-40                                    # f32 denormals
    t["small"] = base * 1e-150
    t["large"] = base * 1e150
    t["one_spot"] = np.array([100.0, 100.0, 180.0, 160.0]) + rng.random((60, 4)) * 1e-3
    z = base.copy()
    z[::4, 2] = z[::4, 0]
    t["zero_width"] = z
    return t


@pytest.mark.parametrize("name", sorted(_tables()))
@pytest.mark.parametrize("thr", [0.98, 0.5, 0.05, 1.0, 1e-12])
@pytest.mark.parametrize("ib", [8, 10])
def test_every_hit_is_inside_the_window(name, thr, ib):
    box = _tables()[name]
    key, limit, y1f, y2f = _records(box, thr, ib)
    hits = _hits(box, thr)
    if name in ("uniform", "integers", "negative") and thr <= 0.98:
        assert hits, "the table should hold hitting pairs"
    for i, j in hits:
        for a, b in ((i, j), (j, i)):                           # whichever of the two comes first in the sorted order
            assert key[b] <= limit[a], (name, thr, i, j)
        assert y2f[i] > y1f[j] and y2f[j] > y1f[i], (name, thr, i, j)


def test_outward_bounds_contain_the_value():
    rng = np.random.default_rng(2)
    v = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.integers(-320, 300, 20000), [0.0, -0.0, 1e-45, -1e-45, 3.5e38, -3.5e38,
                                                                                            1e308, -1e308, np.inf, -np.inf, 16777217.0]])
    lo, hi = _push(v, -1).astype(np.float64), _push(v, +1).astype(np.float64)
    assert np.all(lo <= v) and np.all(hi >= v)
    fin = np.isfinite(v) & (np.abs(v) < 3e38) & (np.abs(v) > 1e-25)       # below, the absolute 2^-120 dominates
    assert np.all((v[fin] - lo[fin]) <= np.abs(v[fin]) * 2.0 ** -21) and np.all((hi[fin] - v[fin]) <= np.abs(v[fin]) * 2.0 ** -21)   # and tight
