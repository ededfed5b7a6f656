#!/usr/bin/env python3
"""Build an A/B variant of librfops.so with extra compiler defines (kernel tuning knobs are
compile-time macros: the product library reads no environment variables).
usage: python tools/build_variant.py TAG -DRFP_SPLIT_BELOW=2048 [-D...]   ->  rfnet_amd/variants/librfops_TAG.so
Load it with RFOPS_LIB=rfnet_amd/variants/librfops_TAG.so (a Python-side switch of rfnet_amd/_lib.py)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from rfnet_amd.build import ARCH, CSRC, HIPCC_FLAGS, _hipcc  # noqa: E402


def main():
    tag, defs = sys.argv[1], sys.argv[2:]
    outdir = os.path.join(ROOT, "rfnet_amd", "variants")
    objdir = os.path.join(outdir, "obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]

    def run(cmd):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise SystemExit(" ".join(cmd) + "\n" + r.stdout)

    with ThreadPoolExecutor(4) as ex:
        list(ex.map(run, [[hipcc] + HIPCC_FLAGS + defs + ["-c", s, "-o", o] for s, o in zip(srcs, objs)]))
    lib = os.path.join(outdir, f"librfops_{tag}.so")
    run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
