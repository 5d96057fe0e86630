"""Build the HIP C-ABI library (gfx950 only) in-tree: mom6_amd/lib/libmom6x.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps device arithmetic
un-fused so that results are bit-comparable with the reference's un-contracted FP64
expression order (all kernels are HBM-bound; FMA contraction buys nothing here).
"""
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmom6x.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = os.environ.get("MOM6X_ARCH", "gfx950")   # (dev A/Bs: gfx950:xnack-; the shipped library is the generic gfx950 code object)
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"] + os.environ.get("MOM6X_CFLAGS", "").split()


# Per-file additions.  continuity_wave.hip: -ffinite-math-only lets the compiler keep compare + select chains shorter (no NaN
# operands to preserve); signed zeros ARE honoured (round 4: -fno-signed-zeros bought nothing -- 2.12 / 2.89 ms per zonal launch
# without it against 2.20 / 3.05 with it, profiles/r04_signed_zero_flags.txt -- and was not where the zeros of opposite sign came
# from: continuity_wave.hip face_column).
PER_FILE = {"continuity_wave.hip": ["-ffinite-math-only"],
            # barotropic.hip: the scheduler's max-ILP strategy takes 4-5 % off k_bt_col (1.41 -> 1.35 ms per launch, four launches per step)
            # and leaves the sub-cycle's kernels where they are; on the other files it is neutral or loses (tracer.hip: +13 % on
            # k_ta_x_tile).  profiles/r06_ab_sched_all.txt
            "barotropic.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _newer(srcs, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


RESOURCES = os.path.join(LIBDIR, "kernel_resources.json")


def _parse_resource_remarks(text):
    """-Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {vgprs, agprs, scratch, occupancy, vgpr_spill, lds}}."""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill", "LDS Size [bytes/block]": "lds"}
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*?): (\d+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1) in keys:
            cur[keys[m.group(1)]] = int(m.group(2))
    return out


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "mom6x.h")]
    objs = []
    # Registers, scratch and occupancy of every kernel as the compiler reports them (kernel_resources.json, next to the library):
    # tests/test_kernel_resources_cpu.py holds the list of kernels that are ALLOWED to spill.  A shared device function that
    # grows can push a hot kernel into scratch memory without any test failing -- round 3 lost 1.6 ms per CorAdCalc call that way.
    try:
        resources = json.load(open(RESOURCES)) if os.path.exists(RESOURCES) else {}
    except Exception:
        resources = {}
    todo = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + ".o")
        base = os.path.basename(s)
        if force or _newer([s] + hdrs, o) or base not in resources:
            todo.append((s, o, base))
        objs.append(o)

    def compile_one(job):
        s, o, base = job
        cmd = [HIPCC] + FLAGS + PER_FILE.get(base, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        return job, cmd, subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    # the files are independent: one hipcc per file, as many at a time as there are cores (MOM6X_BUILD_JOBS overrides)
    jobs = max(1, min(len(todo), int(os.environ.get("MOM6X_BUILD_JOBS", os.cpu_count() or 1))))
    if todo:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            results = list(pool.map(compile_one, sorted(todo, key=lambda j: -os.path.getsize(j[0]))))
        for (s, o, base), cmd, r in results:
            other = "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and
                              not re.match(r"^\s*(\d+ \||\| *\^|\|)", l))
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            if other.strip():
                sys.stderr.write(other + "\n")   # warnings / errors as before
            resources[base] = _parse_resource_remarks(r.stderr)
        json.dump(resources, open(RESOURCES, "w"), indent=0, sort_keys=True)
    if force or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
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
