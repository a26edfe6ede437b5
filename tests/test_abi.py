"""CPU-only: the C-ABI library loads, exports every symbol include/iffnerf_hip.h declares, and fails loudly.

No compute call is made here (there is no GPU in the authoring container).
"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "iffnerf_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(iff_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from iffnerf_amd import _lib, build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/iffnerf_hip.h but not exported"
    # the ctypes table binds exactly the declared surface
    assert sorted(_lib.SIGNATURES) == names
    bound = _lib.lib()
    assert bound.iff_abi_version() == _lib.ABI_VERSION
    assert bound.iff_last_error() == b""


def test_header_cites_the_reference_for_every_entry_point():
    text = open(HEADER).read()
    for name in declared_functions():
        if name in ("iff_last_error", "iff_abi_version", "iff_field_destroy", "iff_idnet_destroy", "iff_vit_destroy", "iff_field_table_bytes",
                    "iff_surface_sample_workspace", "iff_ray_encode_workspace", "iff_q_proj_workspace", "iff_topk_workspace"):
            continue
        pos = text.index(name + "(")
        block = text[max(0, pos - 1500):pos]
        assert re.search(r"\.py:\d+", block), f"{name}: no reference file:line citation above its declaration"


def test_argument_errors_are_reported_without_a_gpu():
    from iffnerf_amd import _lib
    L = _lib.lib()
    assert L.iff_topk(None, 10, 3, None, None, None, 0, None) != 0
    assert b"null" in L.iff_last_error()
    assert L.iff_march_shade(None, None, 6, 4, 0, 20, None, None, None, None, None, None, None, 0, None) != 0
    with pytest.raises(RuntimeError):
        _lib.check(L.iff_attn_logits(None, None, 4, 4, 7, 1.0, None, None, None, 0, None), "iff_attn_logits")
    # the merges of the ray-sharded path: sizes are validated before anything is launched; empty inputs are no-ops
    assert L.iff_merge_row_stats(None, 0, 5, None, None, None) != 0 and b"bad argument" in L.iff_last_error()
    assert L.iff_merge_row_stats(None, 2, 5, None, None, None) != 0 and b"null" in L.iff_last_error()
    assert L.iff_merge_row_stats(None, 2, 0, None, None, None) == 0
    assert L.iff_pack_candidates(None, None, None, None, 0, 3, 101, 100, 0, None, None) != 0          # kl > k
    assert L.iff_pack_candidates(None, None, None, None, 0, 0, 0, 100, 0, None, None) == 0            # no queries
    assert L.iff_merge_candidates(None, 100, 4, 0, 4, 100, None, None, None, None, None) != 0 and b"exceed" in L.iff_last_error()
    # the kept-rows calls: shapes and null buffers are refused before a launch; the row counts are not optional
    assert L.iff_token_assemble_compact(None, 2, 16, 16, 384, None, 0.1, None, None, None, None, None, None) != 0 and b"null" in L.iff_last_error()
    assert L.iff_token_assemble_compact(None, 2, 33, 16, 384, None, 0.1, None, None, None, None, None, None) != 0 and b"bad shape" in L.iff_last_error()
    assert L.iff_token_assemble_compact(None, 0, 16, 16, 384, None, 0.1, None, None, None, None, None, None) == 0
    assert L.iff_logits_from_cache_rows(None, None, 10, None, 256, None, 1.0, None, None, None, None, 0, None) != 0 and b"row counts" in L.iff_last_error()
    # ... and neither is "one 256-row block per image" (only the Python wrapper used to check it)
    counts = (ctypes.c_int32 * 2)(100, 100)
    for m in (100, 300, 0):
        assert L.iff_logits_from_cache_rows(None, None, 10, None, m, ctypes.cast(counts, ctypes.c_void_p), 1.0, None, None, None, None, 0, None) != 0
        assert b"256-row token blocks" in L.iff_last_error(), L.iff_last_error()
    assert L.iff_attn_colsum_rows(None, 2, 256, 10, None, None, None, 0, None, None) != 0 and b"null" in L.iff_last_error()
    assert L.iff_attn_colsum_rows(None, 2, 9000, 10, None, None, None, 0, None, None) != 0 and b"bad shape" in L.iff_last_error()
    assert L.iff_attn_colsum_rows(None, 0, 256, 10, None, None, None, 0, None, None) == 0
    assert L.iff_merge_candidates(None, 8, 4, 2, 3, 100, None, None, None, None, None) != 0            # window past the message
    assert L.iff_merge_candidates(None, 8, 4, 0, 0, 100, None, None, None, None, None) == 0
    assert L.iff_mask_occupied(None, None, 3, None, None) != 0


def test_missing_library_fails_loudly(monkeypatch):
    from iffnerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libiffnerf_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.lib()


def test_cpu_tensors_are_rejected_not_served():
    """There is no CPU path: handing the product CPU tensors raises instead of silently computing on the host."""
    import torch
    from iffnerf_amd import hip_identify as H
    from iffnerf_amd.hip_field import FieldHandle, isocell_emit
    with pytest.raises(RuntimeError, match="no CPU path"):
        H.attn_logits(torch.zeros(2, 16), torch.zeros(3, 16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        H.topk(torch.zeros(8), 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        isocell_emit(torch.zeros(27, 3), torch.zeros(2, 3), torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        FieldHandle.head_only({}, "cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        H.IdNetHandle({}, "cpu")
