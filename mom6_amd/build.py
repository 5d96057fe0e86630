"""Build the HIP C-ABI library (gfx950 only) in-tree: mom6_amd/lib/libmom6x.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps device arithmetic
un-fused so that results are bit-comparable with the reference's un-contracted FP64
expression order (all kernels are HBM-bound; FMA contraction buys nothing here).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmom6x.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"] + os.environ.get("MOM6X_CFLAGS", "").split()


# Per-file additions.  continuity_wave.hip: min/max of the flux limiters and CFL bounds as v_min_f64 / v_max_f64
# instead of compare + two selects (-9 % VALU instructions).  The kernel never produces or tests NaN/Inf, so
# the only observable effect is the SIGN of a zero result (min(-0, +0)); DESIGN.md section 3.
PER_FILE = {"continuity_wave.hip": ["-ffinite-math-only", "-fno-signed-zeros"]}


def _newer(srcs, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "mom6x.h")]
    objs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + ".o")
        if force or _newer([s] + hdrs, o):
            cmd = [HIPCC] + FLAGS + PER_FILE.get(os.path.basename(s), []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        # RCCL is NOT linked: halo.hip resolves ncclSend/ncclRecv/... at run time from the RCCL already in
        # the process (torch's bundled librccl under Python, or the one the Fortran host links), so that a
        # single RCCL instance exists per process.
        cmd += ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
