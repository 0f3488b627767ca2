"""CPU: the operand-range fallback of the facade (an activation beyond the fp16 range must not end a predict call: the
reference runs any checkpoint in fp32) — decision logic and rebuild path with a stub engine; the GPU side is
tests/test_gpu_parity.py::test_facade_falls_back_to_the_bf16_split_mode_when_fp16_overflows."""
import warnings

import pytest

from molnextr_amd import model as M
from molnextr_amd.engine import MNX_ERR_RANGE, MnxError, range_fallback_dtype


def test_fallback_decision():
    rng = MnxError("mnx_predict failed (-6): non-finite features", code=MNX_ERR_RANGE)
    assert range_fallback_dtype(rng, "fp16x3") == "bf16x3"
    assert range_fallback_dtype(rng, "fp16") == "bf16"
    assert range_fallback_dtype(rng, "bf16x3") is None and range_fallback_dtype(rng, "fp32") is None
    assert range_fallback_dtype(MnxError("capacity", code=-5), "fp16x3") is None
    assert range_fallback_dtype(ValueError("x"), "fp16x3") is None


class _StubEngine:
    built = []

    def __init__(self, enc, dec, device=0, max_batch=32, dtype="fp16x3", **kw):
        self.dtype, self.device, self.max_batch, self.closed = dtype, device, max_batch, False
        _StubEngine.built.append(self)

    def close(self):
        self.closed = True


def _facade(monkeypatch, dtype="fp16x3"):
    monkeypatch.setattr(M, "Engine", _StubEngine)
    _StubEngine.built = []
    m = M.molnextr.__new__(M.molnextr)
    m._states, m._max_batch = {"encoder": {}, "decoder": {}}, 8
    m.engine = _StubEngine({}, {}, device=3, max_batch=8, dtype=dtype)
    return m


def test_range_error_rebuilds_once_in_the_bf16_split_mode(monkeypatch):
    m = _facade(monkeypatch)
    first = m.engine
    calls = []

    def job(eng):
        calls.append(eng.dtype)
        if eng.dtype == "fp16x3":
            raise MnxError("mnx_predict failed (-6)", code=MNX_ERR_RANGE)
        return "ok"

    with pytest.warns(RuntimeWarning, match="bf16x3"):
        assert m._with_fallback(job) == "ok"
    assert calls == ["fp16x3", "bf16x3"] and first.closed
    assert m.engine.dtype == "bf16x3" and m.engine.device == 3 and m.engine.max_batch == 8
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert m._with_fallback(job) == "ok"            # the rebuilt engine serves later calls silently


def test_other_errors_and_ranges_without_a_fallback_propagate(monkeypatch):
    m = _facade(monkeypatch)
    with pytest.raises(MnxError):
        m._with_fallback(lambda e: (_ for _ in ()).throw(MnxError("capacity", code=-5)))
    assert len(_StubEngine.built) == 1
    m = _facade(monkeypatch, dtype="bf16x3")
    with pytest.raises(MnxError):
        m._with_fallback(lambda e: (_ for _ in ()).throw(MnxError("range", code=MNX_ERR_RANGE)))
    assert len(_StubEngine.built) == 1
