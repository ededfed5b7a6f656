"""`import rfnet_amd` (and the binding, `rfnet_amd._lib`) must leave the host process's environment
untouched: include/rfops.h promises a library that reads no environment variable, and the package must
not write one behind the host's back (round-2 verdict).  The graph-safe runtime switch is an explicit
opt-in, `rfnet_amd.enable_graph_safe_runtime()`, which the package's own graph-capturing entry points
call.  Also: bare `pytest` from the repository root collects tests/ only."""
import configparser
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
before = dict(os.environ)
import rfnet_amd
from rfnet_amd import _lib, _raw, shard  # the binding and the op layer
import tf_ops.CD.tf_nndistance, pc_distance.tf_approxmatch  # the reference's import paths
assert dict(os.environ) == before, sorted(set(os.environ.items()) ^ set(before.items()))
in_time = rfnet_amd.enable_graph_safe_runtime()
assert os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0" and in_time is True
os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"     # a host's own choice is respected
rfnet_amd.enable_graph_safe_runtime()
assert os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "1"
print("ok")
""" % ROOT


def test_import_leaves_environ_untouched():
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_pytest_ini_limits_collection_to_tests():
    cp = configparser.ConfigParser()
    cp.read(os.path.join(ROOT, "pytest.ini"))
    assert cp["pytest"]["testpaths"].split() == ["tests"]
    # nothing under tools/ may match pytest's default file patterns any more
    bad = [f for d, _, fs in os.walk(os.path.join(ROOT, "tools")) for f in fs
           if f.endswith("_test.py") or (f.startswith("test_") and f.endswith(".py"))]
    assert not bad, bad
