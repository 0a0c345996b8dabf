"""oracle — TEST INFRASTRUCTURE ONLY (CPU checker), never imported by the product.

Two layers:
  * ``oracle.lib``   : ctypes wrappers over ``liboracle.so`` (dyd_oracle.c), the plain-C
                       restatement of the numeric cores at the same array interface as the
                       product's C ABI (include/dyd.h).
  * ``oracle.steps`` : pure-Python / pandas restatement of the five reference step functions
                       (core/processor.py:111-407, 654-831), also timed as bench.py's
                       ``cpu_baseline`` ("port").

Pinning: the reference ships no tests or vectors, so both layers are pinned by
``tests/golden/*.json`` — produced in the build container by ``tests/golden/make_golden.py``,
which imports and runs the reference itself.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import
this package.
"""
