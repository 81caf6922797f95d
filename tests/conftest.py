"""pytest configuration: markers + shared fixtures for the pathpyg_amd test-suite."""
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the reference's own function source (tests/golden/make_golden.py)."""
    return np.load(ROOT / "tests" / "golden" / "reference_vectors.npz", allow_pickle=False)


def golden_delta(golden, name):
    """Rebuild the python/numpy ``delta`` object a golden temporal case was generated with."""
    raw = golden[f"temporal/{name}/delta"]
    kind = str(golden[f"temporal/{name}/delta_kind"])
    if kind == "int":
        return int(raw)
    if kind == "float":
        return float(raw)
    return getattr(np, kind)(raw)
