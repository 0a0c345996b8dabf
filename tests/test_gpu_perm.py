"""K8: numpy's legacy RandomState(seed).permutation(n) computed in parallel on the GPU (csrc/k8_perm.hip), and K6 fed by it
(dyd_split_ids_seeded).  Checked against the reference-generated tests/golden/perm_cases.json, against numpy itself up to
n = 10^8 and against the oracle's sequential loop.  Needs a real MI355X (-m gpu)."""
import time

import numpy as np
import pytest

from conftest import load_golden
from oracle import lib as olib

pytestmark = pytest.mark.gpu


def test_device_permutation_golden(native):
    g = load_golden("perm_cases.json")
    for case in g["perms"]:
        perm, inv = native.mt19937_permutation_device(case["seed"], case["n"], want_inverse=True)
        assert perm.tolist() == case["order"], (case["seed"], case["n"])
        assert np.array_equal(inv[perm], np.arange(case["n"]))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 623, 624, 625, 1249, 4095, 4096, 4097, 65535, 65536, 65537, 1_000_003])
@pytest.mark.parametrize("seed", [0, 42, 2 ** 32 - 1])
def test_device_permutation_matches_numpy(native, n, seed):
    perm, inv = native.mt19937_permutation_device(seed, n, want_inverse=True)
    assert np.array_equal(perm, np.random.RandomState(seed).permutation(n))
    assert np.array_equal(inv[perm], np.arange(n))


def test_device_permutation_hundred_million(native):
    """n = 10^8 (configs[2]'s categories hold 8 * 10^7 records each): identical to numpy, and timed"""
    n, seed = 100_000_000, 42
    native.mt19937_permutation_device(seed, 1000)
    t0 = time.perf_counter()
    perm = native.mt19937_permutation_device(seed, n)
    dt = time.perf_counter() - t0
    want = np.random.RandomState(seed).permutation(n)
    assert np.array_equal(perm, want)
    print(f"\nK8: permutation of {n} in {dt * 1e3:.0f} ms including the 1.6 GB copies back")


@pytest.mark.parametrize("sizes", [[5, 0, 70000, 1, 33000], [200000, 150000], [3]])
def test_split_ids_seeded_matches_the_oracle(native, sizes):
    rng = np.random.default_rng(sum(sizes))
    n_cat = len(sizes)
    cat = np.concatenate([np.full(s, c, np.int32) for c, s in enumerate(sizes)] + [np.full(37, -1, np.int32)])
    rng.shuffle(cat)
    sizes = np.asarray(sizes, np.int64)
    tr, va = (sizes * 0.8).astype(np.int64), (sizes * 0.1).astype(np.int64)
    for seed in (42, 7):
        perm = np.concatenate([olib.mt19937_permutation(seed, int(s)) for s in sizes]) if n_cat else np.zeros(0, np.int64)
        off = np.zeros(n_cat + 1, np.int64)
        np.cumsum(sizes, out=off[1:])
        want = olib.split_ids(cat, perm, off, tr, va)
        got = native.split_ids_seeded(cat, seed, sizes, tr, va)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 1, 1_500_000, (1 << 21) - 1, 1 << 22, 5_000_003, 20_000_000])
@pytest.mark.parametrize("seed", [42, 7, 2 ** 32 - 1])
def test_banded_resolve_equals_the_full_length_rounds(native, n, seed):
    """from 2^20 on K8 resolves the rejections on the short list of count-dependent draws (csrc/k8_perm.hip, k8_classify): same
    permutation as numpy and as the full-length rounds, octave boundaries (2^k) and both sides of them included"""
    L = native.lib()
    want = np.random.RandomState(seed).permutation(n)
    for band in (1, 0):
        native.check(L.dyd_set_option(b"k8_band", band), "opt")
        try:
            perm, inv = native.mt19937_permutation_device(seed, n, want_inverse=True)
        finally:
            native.check(L.dyd_set_option(b"k8_band", 1), "opt")
        assert np.array_equal(perm, want), band
        assert np.array_equal(inv[perm], np.arange(n))
