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
    for sub, mfma in (("k4f_fan_marchILi3", "v_mfma_f32_32x32x2_f32"), ("k4g_fan_marchILi12ELi1ELi3", "v_mfma_f32_32x32x2_f32"),
                      ("k4g_fan_marchILi22ELi3ELi3", "v_mfma_f32_32x32x2_f32"), ("k5_trunk_hILi1ELi1ELi2", "v_mfma_f32_32x32x16_f16"),
                      ("k_vit_gemmILi0ELi128ELi1", "v_mfma_f32_32x32x16_f16"), ("k_vit_gemmILi0ELi128ELi0", "v_mfma_f32_32x32x16_bf16"),
                      ("k_vit_attention_x2", "v_mfma_f32_32x32x16_f16")):
        for n in find(sub):
            assert any(i.startswith(mfma) for i in kernels[n]), (n, mfma)
    for sub in ("k_ss_iter", "k6_colsum", "k7_topk", "k_pose", "k0_mask_cells", "k_mask_occupied", "k_pose_errors"):
        find(sub)
    # the eight-wave fan kernel stages its patches by global -> LDS DMA (no register prefetch, no ds_write pass)
    for n in find("k4g_fan_march"):
        assert sum(1 for i in kernels[n] if i.startswith("global_load_lds_dwordx4")) >= 8, n


def test_product_library_is_the_one_checked(kernels):
    """an IFF_LIB_PATH left exported from an A/B shell must not turn this file into a check of a development build --
    the rules are about the library that ships, so a run under an override fails here instead of passing on the wrong object."""
    from iffnerf_amd import _lib
    assert not _lib.DEV_LIBRARY, f"IFF_LIB_PATH points at {_lib.LIB_PATH}: unset it to check the product library"
