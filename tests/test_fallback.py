"""CPU: the operand-range fallback of the facade (an activation beyond the fp16 range must not end a predict call: the
reference runs any checkpoint in fp32) — decision logic and rebuild path with a stub engine; the GPU side is
tests/test_gpu_parity.py::test_facade_falls_back_to_the_bf16_split_mode_when_fp16_overflows."""
import warnings

import pytest

from molnextr_amd import model as M
from molnextr_amd.engine import DEFAULT_DTYPE, MNX_ERR_RANGE, MnxError, range_fallback_dtype


def test_fallback_decision():
    rng = MnxError("mnx_predict failed (-6): non-finite features", code=MNX_ERR_RANGE)
    assert range_fallback_dtype(rng, "fp16x3") == "bf16x3"
    assert DEFAULT_DTYPE == "fp16x3" and range_fallback_dtype(rng, "fp16x3m") == "bf16x3"
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


def test_a_fallback_after_the_first_group_restarts_the_whole_call(monkeypatch):
    """One predict_images call, one operand mode: when group 2 of 3 trips the range flag, the engine is rebuilt in bf16x3 and
    ALL groups are computed again with it (the groups already done in fp16x3 are not kept)."""
    m = _facade(monkeypatch)
    m.group_images, m.tokenizer, m.device_preprocess = 2, None, False
    seen = []

    def fake_pipeline(eng, x, tok, ref_batch_size=16):
        seen.append((eng.dtype, list(x)))
        if eng.dtype == "fp16x3" and 2 in x:
            raise MnxError("mnx_predict failed (-6)", code=MNX_ERR_RANGE)
        return [{"id": i, "dtype": eng.dtype} for i in x]

    monkeypatch.setattr(M, "predict_pipeline", fake_pipeline)
    monkeypatch.setattr(M.molnextr, "_prefetched", lambda self, groups: iter(groups))
    monkeypatch.setattr(M.molnextr, "_assemble", lambda self, preds, imgs, a, c: preds)
    with pytest.warns(RuntimeWarning, match="bf16x3"):
        out = m.predict_images([0, 1, 2, 3, 4], batch_size=2)
    assert [p["id"] for p in out] == [0, 1, 2, 3, 4] and {p["dtype"] for p in out} == {"bf16x3"}
    assert seen == [("fp16x3", [0, 1]), ("fp16x3", [2, 3]), ("bf16x3", [0, 1]), ("bf16x3", [2, 3]), ("bf16x3", [4])]
    assert m._groups_done == 0


def test_the_prefetch_helper_never_touches_a_closed_engine_nor_shares_one_with_a_second_helper(monkeypatch):
    """the facade uploads + transforms group g + 1 on a helper thread while the engine works on group g.
    When group g trips the range flag, the engine is closed and replaced — the helper may still be inside mnx_preprocess on
    the old handle (use-after-free), or, after the swap, a second helper of the restarted call could run preprocess on the
    new handle beside it. Here with the REAL _prefetched generator (stub engines whose preprocess takes a while and records
    what it sees): no preprocess call may observe its engine closed, no two may overlap on one engine, the abandoned
    generator's helper is joined, and the restarted call still returns every image once, in the fallback mode."""
    import contextlib
    import threading
    import time

    events, lock = [], threading.Lock()

    class _SlowEngine(_StubEngine):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self.busy = 0

        def preprocess(self, images):
            with lock:
                self.busy += 1
                events.append(("enter", id(self), self.closed, self.busy))
            time.sleep(0.05)
            with lock:
                events.append(("exit", id(self), self.closed, self.busy))
                self.busy -= 1
            return list(images)

    monkeypatch.setattr(M, "Engine", _SlowEngine)
    _StubEngine.built = []
    m = M.molnextr.__new__(M.molnextr)
    m._states, m._max_batch = {"encoder": {}, "decoder": {}}, 8
    m.engine = _SlowEngine({}, {}, device=0, max_batch=8, dtype="fp16x3")
    m.group_images, m.tokenizer, m.device_preprocess = 2, None, True
    monkeypatch.setattr(M.molnextr, "_side_context", lambda self: contextlib.nullcontext())
    monkeypatch.setattr(M.molnextr, "_assemble", lambda self, preds, imgs, a, c: preds)

    def fake_pipeline(eng, x, tok, ref_batch_size=16):
        assert not eng.closed
        if eng.dtype == "fp16x3" and 2 in x:
            raise MnxError("mnx_predict failed (-6)", code=MNX_ERR_RANGE)     # while the helper is inside preprocess of group 3
        time.sleep(0.01)
        return [{"id": i, "dtype": eng.dtype} for i in x]

    monkeypatch.setattr(M, "predict_pipeline", fake_pipeline)
    n_threads = threading.active_count()
    with pytest.warns(RuntimeWarning, match="bf16x3"):
        out = m.predict_images(list(range(9)), batch_size=2)
    assert [p["id"] for p in out] == list(range(9)) and {p["dtype"] for p in out} == {"bf16x3"}
    assert all(not closed for (_, _, closed, _) in events), "a preprocess call ran on (or across the close of) a closed engine"
    assert all(busy == 1 for (_, _, _, busy) in events), "two preprocess calls overlapped on one engine handle"
    assert m._prefetch_thread is None and threading.active_count() == n_threads, "a helper thread was left behind"
    assert len(_StubEngine.built) == 2 and _StubEngine.built[0].closed and not _StubEngine.built[1].closed
