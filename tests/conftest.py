"""pytest configuration: markers, import paths, shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI export check (no GPU needed)
`-m gpu`       : parity tests proper -- HIP path through the C-ABI vs oracle / fixtures
"""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def wl():
    return importlib.import_module("lc-crf-slam_amd.workloads")


@pytest.fixture(scope="session")
def po():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def golden():
    return {n: np.load(os.path.join(GOLDEN, n + ".npz"))
            for n in ("slam", "generic", "bilateral", "example_im1", "large")}
