"""CPU: the C-ABI library loads and exports every symbol include/molnextr_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from molnextr_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(engine.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return engine.load_library()


def test_header_symbols_are_exported(lib):
    with open(os.path.join(ROOT, "include", "molnextr_hip.h")) as f:
        hdr = f.read()
    declared = sorted(set(re.findall(r"\b(mnx_[a-z0-9_]+)\s*\(", hdr)))
    assert set(declared) == set(engine.SYMBOLS), (declared, engine.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version(lib):
    assert lib.mnx_abi_version() == engine.ABI_VERSION == 7


def test_config_struct_layout_matches_header():
    # 20 int32 fields + two int32[4] arrays = 26 int32
    assert ctypes.sizeof(engine.MnxConfig) == 26 * 4
    assert ctypes.sizeof(engine.MnxWeightDesc) == 8 + 8 + 8 + 32


def test_create_rejects_bad_arguments_without_gpu(lib):
    h = ctypes.c_void_p()
    assert lib.mnx_create(None, None, 0, 0, ctypes.byref(h)) == -1
    assert b"null" in lib.mnx_last_error(None)
    cfg = engine.MnxConfig()
    desc = (engine.MnxWeightDesc * 1)()
    assert lib.mnx_create(ctypes.byref(cfg), desc, 1, 0, ctypes.byref(h)) == -1   # zeroed config is invalid
    assert b"bad config" in lib.mnx_last_error(None)
    assert not h.value


def _reference_config():
    """include/molnextr_hip.h's mnx_config for the reference model (Swin-B 384, 6-layer decoder), as Engine fills it"""
    cfg = engine.MnxConfig()
    cfg.img_size, cfg.patch, cfg.embed_dim, cfg.n_stages, cfg.window = 384, 4, 128, 4, 12
    for i, (d, h) in enumerate(zip((2, 2, 18, 2), (4, 8, 16, 32))):
        cfg.depths[i], cfg.heads[i] = d, h
    cfg.dec_layers, cfg.dec_dim, cfg.dec_heads, cfg.dec_ff = 6, 256, 8, 1024
    cfg.vocab, cfg.sym_offset, cfg.coord_bins, cfg.pe_len = 256, 128, 64, 5000
    cfg.max_len, cfg.max_batch, cfg.max_atoms, cfg.compute_dtype, cfg.dec_slots = 480, 32, 160, engine.DTYPES["fp16x3"], 2048
    return cfg


def test_create_validates_the_memory_grid_before_touching_a_device(lib):
    """The cross-attention kernels hold <= 512 memory positions (two keys per thread): a config whose last encoder grid is
    larger (768 x 768 pixels: 24 x 24 = 576) must be refused by mnx_create, not decoded with the keys beyond 512 dropped."""
    import torch
    h = ctypes.c_void_p()
    desc = (engine.MnxWeightDesc * 1)()
    cfg = _reference_config()
    rc = lib.mnx_create(ctypes.byref(cfg), desc, 1, 0, ctypes.byref(h))
    if not torch.cuda.is_available():       # the reference config passes the config check: the next refusal is the device
        assert rc != 0 and b"no HIP device" in lib.mnx_last_error(None)
    elif rc == 0:
        lib.mnx_destroy(h)
    cfg.img_size = 768
    h = ctypes.c_void_p()
    assert lib.mnx_create(ctypes.byref(cfg), desc, 1, 0, ctypes.byref(h)) == -1
    msg = lib.mnx_last_error(None)
    assert b"bad config" in msg and b"512" in msg, msg
    assert not h.value


def test_engine_refuses_to_run_without_gpu():
    """No CPU fallback: constructing an Engine on a box without an MI355X must raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from molnextr_amd import weights as W
    tiny = W.EncoderDims(96, 4, 32, (2, 2), (1, 2), 12)
    dec = W.DecoderDims(enc_dim=64)
    ck = W.synthetic_checkpoint(0, enc=tiny, dec=dec)
    with pytest.raises(engine.MnxError, match="no CPU fallback"):
        engine.Engine(ck["encoder"], ck["decoder"], enc=tiny, dec=dec)


def test_header_and_c_example_are_plain_c99(tmp_path):
    """The boundary is a C ABI: the public header and the C usage example must compile with a bare C compiler."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "examples", "predict_c_abi.c")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                        "-c", src, "-o", str(tmp_path / "ex.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
