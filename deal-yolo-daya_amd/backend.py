"""Device-stage backend used by the step functions.

The product backend is ``_native`` (ctypes -> libdyd_gfx950.so -> HIP kernels on gfx950) and
nothing else: ``default_backend()`` raises when the library or the device is missing.  The
step functions take an optional ``backend=`` argument only so that host-side logic
(flatten / emit / sharding) can be exercised without a GPU by the CPU test-suite, which
injects its own checker there; no such object exists inside this package.
"""
from __future__ import annotations

REQUIRED = ("bbox_minmax", "iou_any_ge", "bbox_iou_fused", "hash128", "dedup", "isin", "mt19937_permutation", "split_ids", "yolo_lines")


def default_backend():
    from . import _native

    _native.lib()          # raises NativeUnavailable: no CPU fallback
    return _native


def resolve(backend):
    if backend is None:
        return default_backend()
    missing = [n for n in REQUIRED if not hasattr(backend, n)]
    if missing:
        raise TypeError(f"backend lacks {missing}")
    return backend
