"""Build libposehip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

``python -m sleap_nn_amd.build`` or ``sleap_nn_amd.build.build()``.  hipcc cross-compiles
without a GPU.  The shared library lands in ``sleap_nn_amd/lib/`` (git-ignored, but it does
travel to the GPU box with the working-tree snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libposehip.so")
SOURCES = ["net_kernels.hip", "wino2d_kernels.hip", "wino4_kernels.hip", "w16_kernels.hip", "smallmap_kernels.hip", "f16_kernels.hip", "f16_rows_kernels.hip", "convnext_kernels.hip", "cnblock_mlp_kernels.hip", "model.hip", "train_kernels.hip", "convnext_train_kernels.hip", "train.hip", "post_kernels.hip", "resize_kernels.hip", "group_host.cpp", "comm_rccl.cpp"]
HEADERS = ["common.h", "device_math.h", "act_format.h", "f16_kernels.h", "net_kernels.h", "train_kernels.h", "model_internal.h", os.path.join("..", "..", "include", "posehip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if not force and _newer(obj, [sp] + hdrs):
            continue
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", sp, "-o", obj, "-I", os.path.join(HERE, "..", "include")]
        cmd += os.environ.get("PH_EXTRA_HIPCC_FLAGS", "").split()  # diagnostic builds, e.g. -DPH_STAMP
        if src.endswith(".cpp"):
            cmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", "-x", "c++", "-c", sp, "-o", obj, "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"] if src == "comm_rccl.cpp" else [_hipcc(), "-O3", "-std=c++17", "-fPIC", "-x", "c++", "-c", sp, "-o", obj]
        jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"build failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not _newer(LIB, objs):
        run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
