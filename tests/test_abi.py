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
    assert lib.mnx_abi_version() == engine.ABI_VERSION == 6


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
