"""Disassemble the gfx950 code objects embedded in a HIP shared library / object file (llvm-objdump from the ROCm toolchain).

Test infrastructure: the packed-fp32 rule of tests/test_isa_rules.py reads the instruction stream the library really ships.
"""
import os
import re
import struct
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path: str):
    """[(triple, elf bytes)] of every device code object bundled in ``path`` (one bundle per translation unit)."""
    data = open(path, "rb").read()
    out, at = [], data.find(MAGIC)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", data, at + len(MAGIC))
        off = at + len(MAGIC) + 8
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if sz:
                out.append((triple, data[at + o:at + o + sz]))
        at = data.find(MAGIC, at + len(MAGIC))
    return out


def disassemble(path: str, arch: str = "gfx950"):
    """{kernel symbol: [instruction text, ...]} over all code objects of ``path`` for ``arch``."""
    kernels = {}
    for triple, blob in code_objects(path):
        if arch not in triple:
            continue
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", f"--mcpu={arch}", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                kernels.setdefault(cur, [])
            elif cur is not None and "\t" in line:
                ins = line.split("//")[0].strip()
                if ins:
                    kernels[cur].append(ins)
    return kernels


def available() -> bool:
    return os.path.exists(OBJDUMP)
