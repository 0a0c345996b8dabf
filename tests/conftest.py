import gzip
import io
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def load_golden(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


def golden_csv_text(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read().decode("utf-8-sig")


@pytest.fixture(scope="session")
def oracle_backend():
    from helpers import OracleBackend
    return OracleBackend()


@pytest.fixture(scope="session")
def native():
    """The product's device stage.  Fails loudly (never skips) when the HIP library or GPU is missing."""
    from deal_yolo_daya_amd import _native
    _native.lib()
    return _native
