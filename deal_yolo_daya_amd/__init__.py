"""Importable alias of the package directory ``deal-yolo-daya_amd/``.

The product lives in ``deal-yolo-daya_amd/`` (the name the build contract fixes); a hyphen is
not a legal Python identifier, so this shim makes ``import deal_yolo_daya_amd`` resolve to
that directory: it points the package search path there and runs its ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "deal-yolo-daya_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py"), encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
