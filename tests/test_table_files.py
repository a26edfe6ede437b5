"""Pre-laid-out table files: iff_field_save / _load and iff_idnet_save / _load.

A handle loaded from its file must be the handle it was saved from, bit for bit, on everything the path computes (point
lookups, the march, the surface sampler's draws, logits); files that are not table files, are truncated, or hold the other
kind of handle are refused with RuntimeError -- without touching the GPU where the refusal is decided by the header."""
import os

import numpy as np
import pytest
import torch

from iffnerf_amd import synthetic
from tests import util


def test_bad_files_are_refused_without_a_gpu(tmp_path):
    import ctypes as C
    from iffnerf_amd import _lib
    L = _lib.lib()
    out = C.c_void_p()
    missing = str(tmp_path / "nope.ifft").encode()
    assert L.iff_field_load(missing, None, C.byref(out)) != 0 and b"cannot open" in L.iff_last_error()
    junk = tmp_path / "junk.ifft"
    junk.write_bytes(b"this is not a table file" * 10)
    assert L.iff_field_load(str(junk).encode(), None, C.byref(out)) != 0 and b"IFFTABLE" in L.iff_last_error()
    assert L.iff_idnet_load(str(junk).encode(), None, C.byref(out)) != 0
    with pytest.raises(RuntimeError):
        _lib.check(L.iff_idnet_load(missing, None, C.byref(out)), "iff_idnet_load")
    # a header of the right magic but another version / kind
    hdr = bytearray(64)
    hdr[:8] = b"IFFTABLE"
    hdr[8:12] = (99).to_bytes(4, "little")
    bad = tmp_path / "v99.ifft"
    bad.write_bytes(bytes(hdr))
    assert L.iff_field_load(str(bad).encode(), None, C.byref(out)) != 0 and b"version" in L.iff_last_error()


@pytest.mark.gpu
def test_field_from_file_is_the_same_handle(golden, tmp_path):
    from iffnerf_amd.hip_field import FieldHandle, field_handle_from_ckpt, isocell_emit
    dev = torch.device("cuda:0")
    for which in ("tiny", "small"):
        ck = util.ckpt(which)
        a = field_handle_from_ckpt(ck, dev)
        path = str(tmp_path / f"{which}.ifft")
        a.save(path)
        b = FieldHandle.from_file(path, dev)
        assert b.table_bytes == a.table_bytes and os.path.getsize(path) > a.table_bytes
        assert b.n_samples_default == a.n_samples_default
        x = golden.t("g1_field_points", "xyz").to(dev)
        for fn in ("normalize_coord", "mask_sample", "point_alpha", "point_normals"):
            assert torch.equal(getattr(a, fn)(x), getattr(b, fn)(x)), (which, fn)
        xn = a.normalize_coord(x)
        assert torch.equal(a.density_feature(xn), b.density_feature(xn)) and torch.equal(a.app_feature(xn), b.app_feature(xn))
        rays = golden.t("g2_march_point", "rays").to(dev)
        for mode, S in ((0, 20), (1, -1)):
            ra, rb = a.march(rays, mode, S, want_counts=True), b.march(rays, mode, S, want_counts=True)
            for u, v in zip(ra[:5], rb[:5]):
                assert torch.equal(u, v), (which, mode)
        # the surface sampler reads the occupied-voxel list, which travels in the file's tail
        rho = 0.3
        for u, v in zip(a.surface_sample(75, rho, 4, 200, seed=5), b.surface_sample(75, rho, 4, 200, seed=5)):
            assert torch.equal(u, v)
    # G2 against the golden vectors through the loaded handle alone
    h = FieldHandle.from_file(str(tmp_path / "small.ifft"), dev)
    rgb = h.march(golden.t("g2_march_point", "rays").to(dev), 0, 20)[0]
    torch.testing.assert_close(rgb.cpu(), golden.t("g2_march_point", "rgb"), atol=2e-5, rtol=0)
    # refused: truncated file, and a field file offered as an idnet
    from iffnerf_amd.hip_identify import IdNetHandle
    data = open(tmp_path / "small.ifft", "rb").read()
    (tmp_path / "cut.ifft").write_bytes(data[:len(data) // 2])
    with pytest.raises(RuntimeError, match="truncated"):
        FieldHandle.from_file(str(tmp_path / "cut.ifft"), dev)
    with pytest.raises(RuntimeError):
        IdNetHandle.from_file(str(tmp_path / "small.ifft"), dev)
    with pytest.raises(RuntimeError):
        FieldHandle.from_file(str(tmp_path / "small.ifft"), "cpu")
    # refused: a file whose body does not match what its header / descriptor claim (the loader re-derives the slab layout from
    # the stored dimensions and accepts only exactly that -- no table offset, dimension or list entry is taken on trust)
    import struct

    def corrupt(name, patches, tail=b""):
        b = bytearray(data)
        for off, fmt, val in patches:
            struct.pack_into(fmt, b, off, val)
        (tmp_path / name).write_bytes(bytes(b) + tail)
        return str(tmp_path / name)

    abi, struct_bytes, slab_bytes, n_occ = struct.unpack_from("<IIQQ", data, 16)
    assert len(data) == 64 + struct_bytes + slab_bytes + 4 * n_occ and n_occ > 0
    with pytest.raises(RuntimeError, match="ABI version"):
        FieldHandle.from_file(corrupt("abi.ifft", [(16, "<I", abi + 1)]), dev)
    with pytest.raises(RuntimeError, match="trailing"):
        FieldHandle.from_file(corrupt("tail.ifft", [], tail=b"\0" * 8), dev)
    with pytest.raises(RuntimeError):                                             # a slab size the file cannot hold
        FieldHandle.from_file(corrupt("huge.ifft", [(24, "<Q", 1 << 39)]), dev)
    # the descriptor starts with the 12 table pointers (stored as offset + 1), then basis x 3, mask, cell, head, then grid[3]
    with pytest.raises(RuntimeError, match="do not match"):
        FieldHandle.from_file(corrupt("ptr.ifft", [(64 + 8, "<Q", slab_bytes + 4096)]), dev)
    with pytest.raises(RuntimeError, match="do not match"):
        FieldHandle.from_file(corrupt("ptr2.ifft", [(64, "<Q", 257)]), dev)     # inside the slab, but not where the table is
    grid_off = 64 + 8 * 18
    assert struct.unpack_from("<3i", data, grid_off) == tuple(util.SMALL["grid"])
    with pytest.raises(RuntimeError, match="do not match|out of range"):
        FieldHandle.from_file(corrupt("grid.ifft", [(grid_off, "<i", util.SMALL["grid"][0] + 1)]), dev)
    with pytest.raises(RuntimeError, match="out of range"):
        FieldHandle.from_file(corrupt("grid2.ifft", [(grid_off, "<i", 1 << 20)]), dev)
    with pytest.raises(RuntimeError, match="outside the mask"):
        FieldHandle.from_file(corrupt("occ.ifft", [(len(data) - 4, "<i", 1 << 30)]), dev)


@pytest.mark.gpu
def test_idnet_and_pipeline_from_files(golden, tmp_path):
    from iffnerf_amd import hip_identify as H
    from iffnerf_amd.pipeline import PosePipeline, jitter_scale_from_kwargs
    dev = torch.device("cuda:0")
    w = synthetic.make_id_weights(seed=99)
    o, d, c = (golden.t("g6_identify", k).to(dev) for k in ("ori", "dirs", "rgb"))
    tok = synthetic.make_tokens(256, 384, seed=int(golden["g6_identify"]["tokens_seed"])).to(dev)
    for mode in (H.GEMM_F16X2, H.GEMM_BF16X3, H.GEMM_F32):
        a = H.IdNetHandle(w, dev, gemm_mode=mode)
        path = str(tmp_path / f"id{mode}.ifft")
        a.save(path)
        b = H.IdNetHandle.from_file(path, dev)
        assert (b.feature_c, b.fea, b.img_fea, b.gemm_mode) == (a.feature_c, a.fea, a.img_fea, a.gemm_mode)
        for x, y in zip(a.ray_logits_folded(a.q_fold(tok), o, d, c), b.ray_logits_folded(b.q_fold(tok), o, d, c)):
            assert torch.equal(x, y), mode
        fa, ka = a.ray_encode(o, d, c, want_features=True, want_k=True)
        fb, kb = b.ray_encode(o, d, c, want_features=True, want_k=True)
        assert torch.equal(fa, fb) and torch.equal(ka, kb) and torch.equal(a.q_proj(tok), b.q_proj(tok))
    # a corrupted identification-net file is refused too (width, table offset)
    import struct
    data = bytearray(open(tmp_path / f"id{H.GEMM_F16X2}.ifft", "rb").read())
    bad = bytearray(data); struct.pack_into("<Q", bad, 64, 12345)
    (tmp_path / "idbad.ifft").write_bytes(bytes(bad))
    with pytest.raises(RuntimeError, match="do not match"):
        H.IdNetHandle.from_file(str(tmp_path / "idbad.ifft"), dev)
    # the whole path from two files: same poses, same top-100 as the pipeline built from the checkpoints
    ck = util.ckpt("small")
    p1 = PosePipeline.from_checkpoints(ck, w, dev, model_up=(0.1, 0.2, 0.9))
    p1.save_tables(str(tmp_path / "field.ifft"), str(tmp_path / "idnet.ifft"))
    p2 = PosePipeline.from_table_files(str(tmp_path / "field.ifft"), str(tmp_path / "idnet.ifft"), dev,
                                       jitter_scale_from_kwargs(ck["kwargs"]), model_up=(0.1, 0.2, 0.9))
    r1, r2 = p1.query(tok, 75, seed=3), p2.query(tok, 75, seed=3)
    for x, y in zip(r1, r2):
        assert torch.equal(x, y)
