#!/usr/bin/env python3
"""Build a development library whose ONE translation unit comes from edited device assembly (dev aid; CPU side, no GPU needed).

How the packed-fp32 fault of DESIGN.md section 4 was taken apart in round 4: the device code of a translation unit is compiled to
assembly twice (e.g. with and without the post-RA machine scheduler), split into basic blocks -- the register allocation and the
block structure are identical, only the order inside blocks differs --, recombined block by block or padded with wait states, and
assembled / linked / bundled back into an object exactly as hipcc does it (`hipcc -save-temps -v` shows the steps).

    python scripts/asm_variant.py TU TAG --base BASETAG [--flags "..."] [--alt-flags "..."] [--kernel REGEX]
                                  [--take-alt "mfma" | "not-mfma" | "i,j,k" | "all"] [--nop-after REGEX N] [--nop-before REGEX N]

  TU          translation unit, e.g. fan_march_kernels or trunk_f16_kernels
  TAG         output: build/lib_<TAG>.so (load it with IFF_LIB_PATH)
  --base      the other objects come from build/<BASETAG>/ (made with `python -m iffnerf_amd.build --tag BASETAG -- <flags>`)
  --flags     extra hipcc flags of the primary assembly (default: the faulty configuration, packed fp32 on)
  --alt-flags extra flags of the alternative assembly (default: --flags + "-mllvm -enable-post-misched=0")
  --kernel    regex of the kernel symbol whose blocks are recombined (default: the fused fan march / the fused trunk)
  --take-alt  which basic blocks of that kernel come from the alternative assembly: blocks containing MFMAs, the others, an
              explicit index list, or all
  --nop-after / --nop-before   insert `s_nop N` after / before every instruction matching REGEX (whole translation unit)

Round-4 results with it (truck32k, four graphs in flight, scripts/replay_vs_eager_stages.py ONLY=trunk, 2 400 checked steps each;
"fan" = the fused march with the compiler's own packed fp32 tap combination, "trunk" = k5_trunk_h):
    fan post-scheduled + trunk post-scheduled                      11-21 mismatches        (the fault)
    everything without the post-RA scheduler                        0, 0
    fan post-scheduled, ONLY the trunk without it                   0, 0                    -> the trunk's schedule is the aggressor
    fan without it, trunk post-scheduled                            13, 14                  -> the fan's own schedule does not matter
    trunk built without any packed fp32 instruction, post-scheduled 12, 10                  -> nor do packed instructions in the trunk
    trunk: only its two MFMA blocks unscheduled / only the others   3, 1 / 5, 9             -> mostly the MFMA blocks' issue pattern
    fan padded: s_nop after every packed op / 8 wait states after every LDS wait / before every DPP op     15, 15 / 12, 8 / 8, 12
    fan: blocks of the appearance loops from the unscheduled assembly / all other blocks                    12, 18 / 12, 11
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iffnerf_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DNDEBUG", f"-I{CSRC}"]
PACKED_ON = ["-Xclang", "-target-feature", "-Xclang", "+packed-fp32-ops"]
DEFAULT_KERNEL = {"fan_march_kernels": r"_ZN12_GLOBAL__N_113k4f_fan_marchILi3", "trunk_f16_kernels": r"_Z10k5_trunk_hILi1ELi1ELi2"}


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        sys.exit(f"failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")


def device_asm(tu, flags, out):
    run(["/opt/rocm/bin/hipcc", *BASE_FLAGS, *flags, "--cuda-device-only", "-S", os.path.join(CSRC, tu + ".hip"), "-o", out])
    return open(out).read().split("\n")


def split_kernel(lines, kernel_re):
    start = next(i for i, l in enumerate(lines) if re.match(r"^" + kernel_re + r".*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), [l]]
        else:
            cur[1].append(l)
    blocks.append(cur)
    return start, end, blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tu"); ap.add_argument("tag"); ap.add_argument("--base", required=True)
    ap.add_argument("--flags", default=""); ap.add_argument("--alt-flags", default=None); ap.add_argument("--kernel", default=None)
    ap.add_argument("--take-alt", default=None); ap.add_argument("--nop-after", nargs=2, action="append", default=[])
    ap.add_argument("--nop-before", nargs=2, action="append", default=[])
    a = ap.parse_args()
    work = os.path.join(ROOT, "build", "asm_" + a.tag)
    os.makedirs(work, exist_ok=True)
    flags = PACKED_ON + a.flags.split()
    lines = device_asm(a.tu, flags, os.path.join(work, "primary.s"))
    if a.take_alt:
        alt_flags = (PACKED_ON + a.alt_flags.split()) if a.alt_flags is not None else flags + ["-mllvm", "-enable-post-misched=0"]
        alt = device_asm(a.tu, alt_flags, os.path.join(work, "alt.s"))
        kre = a.kernel or DEFAULT_KERNEL[a.tu]
        s0, e0, b0 = split_kernel(lines, kre)
        _, _, b1 = split_kernel(alt, kre)
        if [b[0] for b in b0] != [b[0] for b in b1]:
            sys.exit("the two assemblies do not have the same basic blocks")
        has_mfma = lambda b: any("v_mfma" in l for l in b[1])   # noqa: E731
        if a.take_alt == "mfma":
            pick = {i for i, b in enumerate(b0) if has_mfma(b)}
        elif a.take_alt == "not-mfma":
            pick = {i for i, b in enumerate(b0) if not has_mfma(b)}
        elif a.take_alt == "all":
            pick = set(range(len(b0)))
        else:
            pick = {int(x) for x in a.take_alt.split(",")}
        body = []
        for i, b in enumerate(b0):
            body += (b1[i][1] if i in pick else b[1])
        lines = lines[:s0 + 1] + body + lines[e0:]
        print(f"{len(pick)} of {len(b0)} blocks of {kre} from the alternative assembly")
    for rx, n in a.nop_after:
        out = []
        for l in lines:
            out.append(l)
            if re.match(rx, l.strip()):
                out.append(f"\ts_nop {int(n)}")
        lines = out
    for rx, n in a.nop_before:
        out = []
        for l in lines:
            if re.match(rx, l.strip()):
                out.append(f"\ts_nop {int(n)}")
            out.append(l)
        lines = out
    dev_s = os.path.join(work, "dev.s")
    open(dev_s, "w").write("\n".join(lines))
    # assemble, link the code object, bundle it, compile the host side around it: the steps of `hipcc -c -save-temps -v`
    run([f"{LLVM}/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-main-file-name", a.tu + ".hip", "-target-cpu", "gfx950",
         "-mrelocation-model", "pic", "-o", os.path.join(work, "dev.o"), dev_s])
    run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-plugin-opt=-amdgpu-internalize-symbols",
         "-plugin-opt=mcpu=gfx950", "--whole-archive", "-o", os.path.join(work, "dev.out"), os.path.join(work, "dev.o"), "--no-whole-archive"])
    run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", f"-input={os.path.join(work, 'dev.out')}", f"-output={os.path.join(work, 'dev.hipfb')}"])
    obj = os.path.join(work, a.tu + ".o")
    run(["/opt/rocm/bin/hipcc", *BASE_FLAGS, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", os.path.join(work, "dev.hipfb"),
         "-c", os.path.join(CSRC, a.tu + ".hip"), "-o", obj])
    base = os.path.join(ROOT, "build", a.base)
    others = [os.path.join(base, f) for f in sorted(os.listdir(base)) if f.endswith(".o") and f != a.tu + ".o"]
    lib = os.path.join(ROOT, "build", f"lib_{a.tag}.so")
    run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *others, obj])
    print(lib)


if __name__ == "__main__":
    main()
