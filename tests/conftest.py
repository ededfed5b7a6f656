"""Shared fixtures.  `-m "not gpu"` runs here on CPU (oracle vs goldens, host logic, ABI
surface, gloo sharding); `-m gpu` runs on the MI355X box and is the parity suite proper:
every GPU test calls the HIP kernels through the C ABI (rfnet_amd._lib -> librfops.so) and
checks them against the CPU oracle (oracle/) -- the oracle is imported ONLY from tests."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # *.so is git-ignored but travels with the repo snapshot: always run the (mtime-incremental)
    # build, so that a source edited after the last build can never be tested against a stale
    # binary (hipcc cross-compiles gfx950 without a GPU; a no-op when everything is current)
    from rfnet_amd.build import build_library
    build_library()


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.oracle import Ref, ref_available, build
    build()
    if not ref_available():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return Ref()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def assert_rel(a, b, rel, abs_=0.0, what=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = abs_ + rel * np.abs(b)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} outside rel={rel} abs={abs_}; max err {err.max():.3e}"
