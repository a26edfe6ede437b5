"""Build libiffnerf_hip.so (hipcc, gfx950 only) in-tree next to the sources.

``python -m iffnerf_amd.build`` or ``iffnerf_amd.build.build()``; hipcc cross-compiles without a GPU.
Objects are rebuilt only when a source or header is newer.  -ffp-contract=off: kernels choose their
own fused multiply-adds, so results do not depend on the compiler's contraction decisions.
-packed-fp32-ops (NO_PACKED_FP32): the compiler may not emit v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32.  Its own packing of fp32
code returned wrong lanes next to MFMA work of another kernel (csrc/fan_march_kernels.hip, lerp_plane_q) and
no operand form explains it, so the rule is the whole instruction class; tests/test_isa_rules.py disassembles the shipped library
and fails on any such instruction.  Measured cost on the headline bench: none.  (The host pass of hipcc does not know the feature
and says so on stderr: harmless.)
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libiffnerf_hip.so")
SOURCES = ["api.hip", "field_kernels.hip", "march_kernels.hip", "fan_march_kernels.hip", "fan8_march_kernels.hip", "march_grad_kernels.hip", "sampler_kernels.hip", "identify_kernels.hip", "trunk_f16_kernels.hip", "vit_kernels.hip",
           "pose_kernels.hip", "shard_kernels.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-DNDEBUG", *NO_PACKED_FP32]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = False, force: bool = False, extra_flags=(), tag: str = "") -> str:
    """``tag``: a development variant -- objects under build/<tag>/, library build/lib_<tag>.so (load it with IFF_LIB_PATH); the
    in-tree product library is left alone."""
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "iffnerf_hip.h"))
    objdir, lib = CSRC, LIB
    if tag:
        objdir = os.path.join(os.path.dirname(HERE), "build", tag)
        lib = os.path.join(os.path.dirname(HERE), "build", f"lib_{tag}.so")
        os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([HIPCC, *FLAGS, *extra_flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        # (the host pass of hipcc does not know the AMDGPU feature named in NO_PACKED_FP32 and says so once per pass: dropped)
        err = "\n".join(l for l in r.stderr.splitlines() if l.strip() and "is not a recognized feature for this target" not in l)
        if verbose and err:
            print(err, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    # python -m iffnerf_amd.build [--force] [--tag NAME -- extra hipcc flags ...]
    argv = sys.argv[1:]
    extra = argv[argv.index("--") + 1:] if "--" in argv else []
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
    print(build(verbose=True, force="--force" in argv, extra_flags=extra, tag=tag))
