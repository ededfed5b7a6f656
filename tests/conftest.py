"""Shared fixtures.  `-m "not gpu"` runs here on CPU (oracle vs goldens, host logic, ABI
surface, gloo sharding); `-m gpu` runs on the MI355X box and is the parity suite proper:
every GPU test calls the HIP kernels through the C ABI (rfnet_amd._lib -> librfops.so) and
checks them against the CPU oracle (oracle/) -- the oracle is imported ONLY from tests."""
import os
import sys

# Before anything starts the HIP runtime (pytest_collection_modifyitems below asks torch for a device):
# captured graphs that hold torch reductions replay stale results on ROCm 7 without it
# (rfnet_amd/_lib.py).
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np  # noqa: E402
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


def strict_bar_report(what, got, exp, rel=1e-4, abs_=1e-6):
    """SURVEY.md 8(d)'s bar for `match` entries is abs 1e-6 + rel 1e-4 on EVERY entry.  Where a test
    accepts less (large clouds: the ten-level annealing amplifies 1-ulp exp differences)
    the strict bar is still evaluated and REPORTED -- printed, attached to the junit record, and raised
    as a warning when not every entry passes, so it shows in the pytest summary -- so that a
    regression of the pass fraction is visible even though the test's own threshold is looser."""
    import warnings
    got = np.asarray(got, np.float64)
    exp = np.asarray(exp, np.float64)
    err = np.abs(got - exp)
    inside = err <= abs_ + rel * np.abs(exp)
    frac, worst = float(inside.mean()), float(err.max())
    msg = (f"strict bar (abs {abs_:g} + rel {rel:g}) {what}: {frac * 100:.5f} % of {got.size} entries inside, "
           f"{int((~inside).sum())} outside, max abs err {worst:.3e}")
    print(msg)
    if frac < 1.0:
        warnings.warn(msg)
    return frac, worst
