"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/dyd.h
declares, with a ctypes prototype for each (no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from deal_yolo_daya_amd import _native

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dyd.h")


def declared_symbols():
    text = open(HEADER, encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dyd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_contract():
    syms = declared_symbols()
    for must in ("dyd_init", "dyd_shutdown", "dyd_last_error", "dyd_device_count", "dyd_bbox_minmax",
                 "dyd_iou_any_ge", "dyd_hash128", "dyd_dedup", "dyd_isin", "dyd_mt19937_permutation",
                 "dyd_split_ids", "dyd_malloc", "dyd_free", "dyd_h2d", "dyd_d2h", "dyd_last_kernel_ms",
                 "dyd_bbox_minmax_dev", "dyd_iou_any_ge_dev", "dyd_hash128_dev", "dyd_dedup_dev", "dyd_isin_dev",
                 "dyd_split_ids_dev", "dyd_dedup_global_dev", "dyd_bbox_iou_fused_dev"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()          # dlopen only; no device needed
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/dyd.h but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes prototype in _native.SIGNATURES"
    assert set(_native.SIGNATURES) == set(declared_symbols())


def test_host_only_entry_points():
    """dyd_version / dyd_last_error / dyd_mt19937_permutation run without a device."""
    lib = _native.load_library()
    assert lib.dyd_version().decode().startswith("dyd ")
    assert _native.mt19937_permutation(42, 10).tolist() == np.random.RandomState(42).permutation(10).tolist()
    assert _native.mt19937_permutation(0, 0).tolist() == []
    for seed, n in ((0, 1), (1, 2), (42, 1000), (4294967295, 257), (9999, 65536 + 3)):
        assert np.array_equal(_native.mt19937_permutation(seed, n), np.random.RandomState(seed).permutation(n))
    with pytest.raises(ValueError):
        _native.mt19937_permutation(2 ** 32, 3)
    assert lib.dyd_mt19937_permutation(1, -1, None) == -1
    assert b"invalid" in lib.dyd_last_error()
