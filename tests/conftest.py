import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def scan_map():
    return (np.load(os.path.join(GOLDEN, "points_scan.npy")),
            np.load(os.path.join(GOLDEN, "points_map.npy")))


@pytest.fixture(autouse=True)
def _certificates_at_every_size(monkeypatch):
    """The product uses match certificates only from ~2 M certified point-iterations on (dicp_amd._ops.CERT_MIN_WORK: below that they cost the host more
    than they save the GPU).  The parity tests are about what the certified loop COMPUTES: they run it at every size, as before the policy existed
    (tests/test_gpu_configs.py::test_certificates_are_used_where_they_pay checks the policy itself)."""
    from dicp_amd import _ops
    monkeypatch.setattr(_ops, "CERT_MIN_WORK", 0.0)
