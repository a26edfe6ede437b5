"""CPU-only: rules on the instruction stream the library ships (llvm-objdump of the gfx950 code objects inside the .so).

Rule 1 -- no packed fp32 arithmetic.  The compiler's own v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 code returned wrong lanes in the
fan march while an MFMA kernel shared the CU (csrc/fan_march_kernels.hip, lerp_plane_q): the library is built
with -packed-fp32-ops (iffnerf_amd/build.py) and this test holds the line for every kernel of every translation unit.
"""
import os
import re

import pytest

from tests import isa_util

PACKED_FP32 = re.compile(r"^v_pk_(mul|add|fma)_f32\b")


@pytest.fixture(scope="module")
def kernels():
    if not isa_util.available():
        pytest.skip("llvm-objdump of the ROCm toolchain is not installed")
    from iffnerf_amd import _lib, build
    build.build()
    k = isa_util.disassemble(_lib.LIB_PATH)
    assert len(k) >= 80 and sum(len(v) for v in k.values()) > 100000, "the disassembly looks empty"
    return k


def test_no_packed_fp32_instruction_in_the_library(kernels):
    bad = {name: sum(1 for i in ins if PACKED_FP32.match(i)) for name, ins in kernels.items()}
    bad = {n: c for n, c in bad.items() if c}
    assert not bad, f"packed fp32 instructions in {len(bad)} kernels, e.g. {sorted(bad.items(), key=lambda t: -t[1])[:5]}"


def test_the_hot_kernels_are_in_the_library_and_use_the_matrix_cores(kernels):
    """The census doubles as a build check: the kernels DESIGN.md names exist, and the ones that should issue MFMAs do."""
    def find(sub):
        hits = [n for n in kernels if sub in n]
        assert hits, f"no kernel named *{sub}*"
        return hits
    for sub, mfma in (("k4f_fan_marchILi3", "v_mfma_f32_32x32x2_f32"), ("k5_trunk_hILi1ELi1ELi2", "v_mfma_f32_32x32x16_f16"),
                      ("k_vit_gemmILi0ELi128ELi1", "v_mfma_f32_32x32x16_f16"), ("k_vit_gemmILi0ELi128ELi0", "v_mfma_f32_32x32x16_bf16"),
                      ("k_vit_attention_x2", "v_mfma_f32_32x32x16_f16")):
        for n in find(sub):
            assert any(i.startswith(mfma) for i in kernels[n]), (n, mfma)
    for sub in ("k_ss_iter", "k6_colsum", "k7_topk", "k_pose", "k0_mask_cells", "k_mask_occupied"):
        find(sub)


def test_experiment_patches_still_apply():
    """scripts/experiments/*.patch are measured negatives kept reproducible: each must apply to the kernel sources
    as committed (`git apply --check` reads the patch and the tree, it needs no repository)."""
    import glob
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("git") is None:
        pytest.skip("no git")
    patches = sorted(glob.glob(os.path.join(root, "scripts", "experiments", "*.patch")))
    assert patches
    for p in patches:
        r = subprocess.run(["git", "apply", "--check", p], cwd=root, capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stderr)
