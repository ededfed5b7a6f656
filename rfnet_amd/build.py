"""Builds rfnet_amd/librfops.so from csrc/*.hip with hipcc for gfx950 (in-tree, no JIT cache).

hipcc cross-compiles without a GPU, so this runs in the build container and the resulting
.so travels to the GPU box with the repo snapshot.  `python -m rfnet_amd.build [--force]`.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "build")
LIB = os.path.join(PKG, "librfops.so")
ARCH = "gfx950"

# -ffp-contract=off: every FMA in the kernels is an explicit fmaf(), so the fp32 instruction
# sequence is pinned (parity with the reference CUDA ops).
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    # packed fp32 (v_pk_*) is no faster than scalar fp32 on gfx950 VALU (tools/ubench/valu_rate.hip)
    # and SLP-vectorising into it costs operand shuffles
    "-fno-slp-vectorize",
    "-Wall",
    "-Wno-unused-function",
]
# per-kernel register / scratch / LDS figures of every compile, kept next to the objects (csrc/build/*.resources.txt) and read
# by kernel_resources(): tests/test_kernel_resources.py holds the hot kernels to "no scratch, the occupancy they were tuned for"
# (round 4 found a re-scan batch and a histogram loop that spilled without anyone noticing)
RESOURCE_FLAG = "-Rpass-analysis=kernel-resource-usage"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [
        os.path.join(PKG, "..", "include", "rfops.h")
    ]
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs) or not os.path.exists(o[:-2] + ".resources.txt"):
            jobs.append([hipcc] + HIPCC_FLAGS + [RESOURCE_FLAG, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for cmd, out in zip(jobs, ex.map(run, jobs)):
                remarks = [ln for ln in out.splitlines() if "kernel-resource-usage" in ln]
                with open(cmd[-1][:-2] + ".resources.txt", "w") as f:
                    f.write("\n".join(remarks) + "\n")
                rest = "\n".join(ln for ln in out.splitlines() if "kernel-resource-usage" not in ln and not ln.startswith(("   ", "      |")))
                if verbose and rest.strip():
                    print(rest)
    if force or jobs or _newer(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def kernel_resources():
    """{mangled kernel name: {"sgprs", "vgprs", "agprs", "scratch", "occupancy", "sgpr_spill", "vgpr_spill", "lds"}} of the
    library as last compiled (hipcc's kernel-resource-usage remarks; builds the library if a report is missing)."""
    build_library()
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "LDS Size [bytes/block]": "lds"}
    res, cur = {}, None
    for path in sorted(glob.glob(os.path.join(OBJ, "*.resources.txt"))):
        for ln in open(path):
            body = ln.split("remark:", 1)[-1].rsplit("[-Rpass", 1)[0].strip()
            if body.startswith("Function Name:"):
                cur = res.setdefault(body.split(":", 1)[1].strip(), {})
            elif cur is not None and ":" in body:
                k, v = body.rsplit(":", 1)
                if k.strip() in keys and v.strip().lstrip("-").isdigit():
                    cur[keys[k.strip()]] = int(v)
    return res


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose=True)
    print("built", path)
