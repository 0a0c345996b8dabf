"""MI355X-native (gfx950) annotation hot path of Cyclones-Y/Deal-Yolo-Daya.

Drop-in for five step functions of the reference's ``core/processor.py`` (same names,
signatures, return values and error behaviour — see ``core/processor.py`` here), with the
numeric cores running as hand-written HIP kernels in ``csrc/`` behind the C ABI of
``include/dyd.h`` (loaded through ctypes by ``_native``).  There is no CPU fallback: if
``libdyd_gfx950.so`` or a gfx950 device is missing, the step functions raise.
"""
__version__ = "0.1.0"
