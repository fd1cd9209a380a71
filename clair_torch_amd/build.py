"""Build the gfx950 HIP library (clair_torch_amd/lib/libclair_hip.so) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build container.
-ffp-contract=off: several float32 expressions must round exactly as the reference's un-fused eager
PyTorch ops do; the fused multiply-adds the kernels want are spelled __builtin_fmaf explicitly.
-fno-slp-vectorize: packed float32 VALU ops (v_pk_*_f32) that SLP forms are slower than the scalar ops
they replace on gfx950 for these kernels (measured 10 % on the merge kernel, tools/merge_variants.hip).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libclair_hip.so")
OBJ_DIR = os.path.join(LIB_DIR, "obj")   # per-source objects (git-ignored), so one edited kernel recompiles alone
SOURCES = ["ct_merge.hip", "ct_merge_multi.hip", "ct_merge_exact.hip", "ct_linearize.hip", "ct_pairs.hip", "ct_flatfield.hip", "ct_stats.hip", "ct_darkfield.hip", "ct_bandstats.hip",
           "ct_api.cpp"]
HEADERS = ["ct_device.hpp", "ct_merge.hpp", os.path.join("..", "..", "include", "clair_hip.h")]
EXTRA_DEPS = {"ct_merge_multi.hip": ["ct_merge.hip"]}  # sources that include another source verbatim
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps += [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > built for d in deps)


def _object_is_stale(src, obj):
    if not os.path.exists(obj):
        return True
    built = os.path.getmtime(obj)
    extra = [os.path.join(CSRC, e) for e in EXTRA_DEPS.get(os.path.basename(src), [])]
    return any(os.path.getmtime(d) > built for d in [src] + extra + [os.path.join(CSRC, h) for h in HEADERS])


def build(force=False, verbose=False):
    """Compile every HIP source (in parallel, one object each) and link them into one shared library."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = [os.path.join(OBJ_DIR, os.path.basename(s) + ".o") for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not force and not _object_is_stale(src, obj):
            return
        cmd = [hipcc] + FLAGS + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, zip(srcs, objs)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
