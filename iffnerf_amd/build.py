"""Build libiffnerf_hip.so (hipcc, gfx950 only) in-tree next to the sources.

``python -m iffnerf_amd.build`` or ``iffnerf_amd.build.build()``; hipcc cross-compiles without a GPU.
Objects are rebuilt only when a source or header is newer.  -ffp-contract=off: kernels choose their
own fused multiply-adds, so results do not depend on the compiler's contraction decisions.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libiffnerf_hip.so")
SOURCES = ["api.hip", "field_kernels.hip", "march_kernels.hip", "fan_march_kernels.hip", "march_grad_kernels.hip", "sampler_kernels.hip", "identify_kernels.hip", "trunk_f16_kernels.hip", "vit_kernels.hip",
           "pose_kernels.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-DNDEBUG"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = False, force: bool = False, extra_flags=()) -> str:
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "iffnerf_hip.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([HIPCC, *FLAGS, *extra_flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
