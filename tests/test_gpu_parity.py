"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) the golden vectors generated from the
reference and (b) the CPU oracle on the same seeded inputs.

Tolerances (stated per BASELINE north_star "within 1e-3 on logits, exact on predicted SMILES/atom-bond sets"):
  * decoder / bond head are fp32 on the GPU: logits, log-probs, hidden  <= 1e-3 abs (measured ~3e-5); token ids,
    lengths and bond classes EXACT.
  * encoder: the module-wide engine runs the DEFAULT operand mode, fp16x3 (split fp16 operands, three MFMA terms per
    product, fp32 accumulate): features <= 5e-5 abs on unit-variance features (measured 5e-6). The plain 16-bit modes
    are exercised per kernel and per block: bf16 <= 4e-2 .. 6e-2, fp16 <= 6e-3 .. 1e-2 (tests/test_gpu_pixels.py carries
    the from-pixels figures of every mode).
"""
import json
import os

import numpy as np
import pytest
import torch

from molnextr_amd import weights as W

pytestmark = pytest.mark.gpu

TINY = W.EncoderDims(img_size=96, patch=4, embed_dim=32, depths=(2, 2), heads=(1, 2), window=12)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def eng(synth_ckpt, dev):
    from molnextr_amd.engine import Engine
    e = Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=32, dtype="fp16x3")
    yield e
    e.close()


def _tiny_engine(dtype):
    from molnextr_amd.engine import Engine
    dec = W.DecoderDims(enc_dim=TINY.num_features)
    ck = W.synthetic_checkpoint(0, enc=TINY, dec=dec)
    return Engine(ck["encoder"], ck["decoder"], max_batch=2, enc=TINY, dec=dec, dtype=dtype)


def test_native_library_is_loaded(eng):
    """The driver records which in-tree .so the test process loaded: make sure it is ours."""
    with open("/proc/self/maps") as f:
        assert "libmolnextr_hip.so" in f.read()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 96, 32), (129, 136, 72), (4608, 1024, 4096), (18432, 2048, 512),
                                   (1000, 384, 128), (640, 768, 256), (2304, 128, 512), (515, 64, 192)])
def test_gemm_all_epilogues(dev, M, N, K):
    eng = _tiny_engine("bf16")            # mnx_gemm16 runs the engine's 16-bit operand type
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()   # asymmetric, transposition-detecting
    bias = torch.randn(N, generator=g).to(dev)
    ref = A.float() @ Wt.float().t() + bias
    out = torch.zeros(M, N, device=dev)
    eng.gemm16(3, A, Wt, out, bias)
    assert (out - ref).abs().max().item() < 2e-4
    o16 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    eng.gemm16(0, A, Wt, o16, bias)
    assert (o16.float() - ref).abs().max().item() < 4e-2
    eng.gemm16(1, A, Wt, o16, bias)
    assert (o16.float() - torch.nn.functional.gelu(ref)).abs().max().item() < 4e-2
    res = torch.randn(M, N, generator=g).to(dev)
    r2 = res.clone()
    eng.gemm16(2, A, Wt, r2, bias)
    assert (r2 - (ref + res)).abs().max().item() < 2e-4
    eng.close()


def _split_planes(x, td, scale=1.0):
    hi = (x * scale).to(td)
    lo = (x * scale - hi.float()).to(td)
    return torch.stack([hi, lo]).contiguous()


@pytest.mark.parametrize("dtype", ["fp16x3", "bf16x3"])
@pytest.mark.parametrize("M,N,K", [(300, 96, 32), (129, 136, 72), (1000, 384, 128), (2304, 128, 512), (4608, 1024, 4096),
                                   (16384, 1024, 256), (18432, 2048, 512), (65536, 256, 128)])
def test_gemm_split_operand_modes(dev, dtype, M, N, K):
    """The split-operand GEMM (hi.hi + hi.lo + lo.hi on the 16-bit MFMA) of both kernels — 128-tile (DMA and non-DMA K
    loops) and the six-phase persistent 256x256 one (gemm256x3_kernel: the last three shapes, all four epilogues; the
    576-tile shape splits into 2 whole rounds on it + the remaining rows on the 128-tile kernel) — against a float64 product of the
    SAME fp32 inputs, every output element: fp32-class accuracy from 16-bit matrix instructions, the exact-erf GELU, the
    hi + lo output planes, the power-of-two weight scale, and terms = 1 degrading to the plain 16-bit product."""
    e = _tiny_engine(dtype)
    td = torch.float16 if dtype == "fp16x3" else torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = (A.double() @ Wt.double().t() + bias.double())
    wscale = 2.0 ** 12 if dtype == "fp16x3" else 1.0
    A2, W2 = _split_planes(A, td), _split_planes(Wt, td, wscale)
    # what the planes can represent at best: the product of the rounded sums
    tol = 3e-6 if dtype == "fp16x3" else 2e-4
    out = torch.zeros(M, N, device=dev)
    e.gemm16_split(3, A2, W2, out, bias, oscale=1.0 / wscale)
    assert (out.double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    out.fill_(7.0)
    e.gemm16_split(3, A2, W2, out, None, oscale=1.0 / wscale)          # a layer without bias (patch-merging reduction)
    assert (out.double() - (ref - bias.double())).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    res = torch.randn(M, N, generator=g).to(dev)
    r2 = res.clone()
    e.gemm16_split(2, A2, W2, r2, bias, oscale=1.0 / wscale)
    assert (r2.double() - (ref + res.double())).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    again = res.clone()
    e.gemm16_split(2, A2, W2, again, bias, oscale=1.0 / wscale)
    assert torch.equal(again, r2)                                       # same launch twice: bit-identical
    for epi, want in ((0, ref), (1, torch.nn.functional.gelu(ref))):
        o2 = torch.zeros(2, M, N, device=dev, dtype=td)
        e.gemm16_split(epi, A2, W2, o2, bias, oscale=1.0 / wscale)
        got = o2[0].double() + o2[1].double()
        assert (got - want).abs().max().item() < tol * max(1.0, want.abs().max().item()), epi
        half_ulp = 2.0 ** -10 if td == torch.float16 else 2.0 ** -7          # |lo| <= half an ulp of hi
        assert bool((o2[1].float().abs() <= o2[0].float().abs() * half_ulp + 1e-7).all()), "lo must be hi's rounding residual"
    o1 = torch.zeros(2, M, N, device=dev, dtype=td)
    e.gemm16_split(0, A2, W2, o1, bias, oscale=1.0 / wscale, terms=1)
    plain = A2[0].double() @ W2[0].double().t() / wscale + bias.double()
    assert ((o1[0].double() + o1[1].double()) - plain).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    e.close()


@pytest.mark.parametrize("M,N,K", [(300, 96, 32), (129, 136, 72), (1000, 384, 128), (2304, 128, 512), (4608, 1024, 4096),
                                   (16384, 1024, 256), (18432, 2048, 512), (65536, 256, 128), (65536, 512, 128)])
def test_gemm_two_term_form(dev, M, N, K):
    """SplitArgs::terms == 2 (compute_dtype FP16X3M: C = oscale (Ah.Wh + Ah.Wl) + bias, the activation's lo plane neither read
    nor multiplied) on both kernels — the 128-tile one (DMA and non-DMA K loops) and the four-phase form of the persistent
    256x256 kernel (the last five shapes; one, two and several K-tiles per tile, several tiles per workgroup, a ragged round)
    — against a float64 product of the ROUNDED activation (the hi plane) with the unrounded weight, every output element,
    all four epilogues; the one-plane output (hi only, for a two-term consumer) is the two-plane output's hi plane bit for bit;
    and the two kernels add the same fp32 numbers in the same order (an image's features must not depend on which kernel
    its rows land in: launch_gemm16 splits a layer's rows between them by batch size)."""
    e = _tiny_engine("fp16x3")
    td = torch.float16
    g = torch.Generator().manual_seed(M + N + K + 2)
    A = torch.randn(M, K, generator=g).to(dev)
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    wscale = 2.0 ** 12
    A2, W2 = _split_planes(A, td), _split_planes(Wt, td, wscale)
    A2[1].fill_(float("nan"))                                            # the lo plane of A must not be touched
    ref = A2[0].double() @ Wt.double().t() + bias.double()
    tol = 3e-6
    out = torch.zeros(M, N, device=dev)
    e.gemm16_split(3, A2, W2, out, bias, oscale=1.0 / wscale, terms=2)
    assert (out.double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    o128 = torch.zeros(M, N, device=dev)
    e.gemm16_split(3 | 0x200, A2, W2, o128, bias, oscale=1.0 / wscale, terms=2)
    assert torch.equal(o128, out), "four-phase 256x256 kernel and 128x128 kernel must agree bit for bit"
    full = A.double() @ Wt.double().t() + bias.double()
    assert (out.double() - full).abs().max().item() > 20 * tol, "the activation's rounding must be visible (else lo was read)"
    res = torch.randn(M, N, generator=g).to(dev)
    r2 = res.clone()
    e.gemm16_split(2, A2, W2, r2, bias, oscale=1.0 / wscale, terms=2)
    assert (r2.double() - (ref + res.double())).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    r128 = res.clone()
    e.gemm16_split(2 | 0x200, A2, W2, r128, bias, oscale=1.0 / wscale, terms=2)
    assert torch.equal(r128, r2)
    for epi, want in ((0, ref), (1, torch.nn.functional.gelu(ref))):
        o2 = torch.zeros(2, M, N, device=dev, dtype=td)
        e.gemm16_split(epi, A2, W2, o2, bias, oscale=1.0 / wscale, terms=2)
        got = o2[0].double() + o2[1].double()
        assert (got - want).abs().max().item() < tol * max(1.0, want.abs().max().item()), epi
        o128 = torch.zeros(2, M, N, device=dev, dtype=td)
        e.gemm16_split(epi | 0x200, A2, W2, o128, bias, oscale=1.0 / wscale, terms=2)
        assert torch.equal(o128, o2), epi
        if epi == 1:        # the GELU epilogue with one output plane (fc1 when fc2 runs on two terms), both term counts
            for terms in (2, 3):
                a_in = A2 if terms == 2 else _split_planes(A, td)
                two = torch.zeros(2, M, N, device=dev, dtype=td)
                e.gemm16_split(1, a_in, W2, two, bias, oscale=1.0 / wscale, terms=terms)
                one = torch.full((1, M, N), 3.0, device=dev, dtype=td)
                e.gemm16_split(1 | 0x400, a_in, W2, one, bias, oscale=1.0 / wscale, terms=terms)
                assert torch.equal(one[0], two[0]), terms
                one128 = torch.full((1, M, N), 3.0, device=dev, dtype=td)
                e.gemm16_split(1 | 0x400 | 0x200, a_in, W2, one128, bias, oscale=1.0 / wscale, terms=terms)
                assert torch.equal(one128[0], two[0]), terms
    e.close()
    eb = _tiny_engine("bf16x3")
    with pytest.raises(Exception):          # bf16 operands never run on two terms
        eb.gemm16_split(3, A2.to(torch.bfloat16), W2.to(torch.bfloat16), out, bias, terms=2)
    eb.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(65536, 256, 128), (16384, 1024, 256), (18432, 2048, 512), (9216, 3072, 1024)])
def test_gemm_persistent_256_tile_kernel(dev, dtype, M, N, K):
    """Shapes that launch_gemm16 routes to gemm256.hip (M, N multiples of 256, >= 256 tiles, 16-bit-output epilogues):
    one round (256 tiles), the two-K-tile case (K = 128), several rounds with a ragged last one; bias and bias + GELU,
    both 16-bit types; every output element is compared (tile seams, first / last tile of every workgroup)."""
    e = _tiny_engine(dtype)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev).to(td)
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(td)
    bias = torch.randn(N, generator=g).to(dev)
    ref = A.float() @ Wt.float().t() + bias
    tol = 4e-2 if dtype == "bf16" else 6e-3
    o16 = torch.zeros(M, N, device=dev, dtype=td)
    e.gemm16(0, A, Wt, o16, bias)
    assert (o16.float() - ref).abs().max().item() < tol
    o16.zero_()
    e.gemm16(1, A, Wt, o16, bias)
    assert (o16.float() - torch.nn.functional.gelu(ref)).abs().max().item() < tol
    again = torch.zeros_like(o16)
    e.gemm16(1, A, Wt, again, bias)
    assert torch.equal(again, o16)                       # same launch twice: bit-identical
    e.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp16x3"])
@pytest.mark.parametrize("M,N,K", [(65536, 128, 128), (65536, 128, 512), (16384, 512, 2048), (73728, 512, 512),
                                   (32768, 256, 256), (18432, 1024, 1024)])
def test_gemm_persistent_residual_kernel(dev, dtype, M, N, K):
    """gemm_res.hip, the persistent 256x128 kernel for the fp32-output epilogues (M % 256 == 0, N % 128 == 0), forced
    through the test-aid bit of mnx_gemm16 (the dispatch prefers it for plain 16-bit operands at >= 1024 tiles only):
    the two-K-tile case (K = 128: the residual prefetch is issued in a tile's FIRST K-tile), exactly one round (256 tiles),
    ragged rounds (4.5, 2.25), long K; bias + in-place fp32 residual, and bias -> fp32 without residual and without bias;
    plain 16-bit and split operands; every output element is compared; the same launch twice is bit-identical."""
    e = _tiny_engine(dtype)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    if dtype == "fp16x3":
        ws = 2.0 ** 12
        A2, W2 = _split_planes(A, torch.float16), _split_planes(Wt, torch.float16, ws)
        ref = A.double() @ Wt.double().t()
        run = lambda epi, out, b: e.gemm16_split(epi | 0x100, A2, W2, out, b, oscale=1.0 / ws)   # noqa: E731
        tol = 3e-6 * 8
    else:
        td = torch.bfloat16 if dtype == "bf16" else torch.float16
        A16, W16 = A.to(td), Wt.to(td)
        ref = A16.double() @ W16.double().t()
        run = lambda epi, out, b: e.gemm16(epi | 0x100, A16, W16, out, b)                # noqa: E731
        tol = 2e-4
    r2 = res.clone()
    run(2, r2, bias)
    assert (r2.double() - (ref + bias.double() + res.double())).abs().max().item() < tol
    again = res.clone()
    run(2, again, bias)
    assert torch.equal(again, r2)
    out = torch.full((M, N), 7.0, device=dev)
    run(3, out, bias)
    assert (out.double() - (ref + bias.double())).abs().max().item() < tol
    out.fill_(7.0)
    run(3, out, None)                                    # the patch-merging reduction has no bias
    assert (out.double() - ref).abs().max().item() < tol
    e.close()


@pytest.mark.parametrize("dtype,tol", [("bf16", 4e-2), ("fp16", 6e-3), ("bf16x3", 2e-4), ("fp16x3", 2e-5), ("fp32", 2e-5)])
def test_swin_tiny_every_block_vs_reference_golden(golden_dir, dev, dtype, tol):
    gold = np.load(os.path.join(golden_dir, "swin_tiny.npz"))
    e = _tiny_engine(dtype)
    img = W.hash_normal("swin_tiny_img", (2, 3, 96, 96), 1.0).to(dev)
    for i, name in enumerate(["patch_embed", "s0b0", "s0b1", "merge0", "s1b0", "s1b1"]):
        dst = torch.zeros(gold[name].shape, device=dev)
        e.set_tap(i, dst)
        e.encode(img)
        torch.cuda.synchronize()
        err = np.abs(dst.cpu().numpy() - gold[name]).max()
        assert err < (1e-5 if name == "patch_embed" else tol), (name, err)
    e.set_tap(-1, None)
    assert np.abs(e.encode(img).cpu().numpy() - gold["features"]).max() < tol
    e.close()


def test_swin_full_vs_reference_golden(golden_dir, eng, dev):
    gold = np.load(os.path.join(golden_dir, "swin_full.npz"))
    f = eng.encode(W.synthetic_images(2).to(dev)).cpu().numpy()
    assert f.shape == (2, 144, 1024)
    assert np.abs(f[:, :4, :] - gold["features_head"]).max() < 5e-5
    assert np.abs(f[:, ::9, ::16] - gold["features_strided"]).max() < 5e-5
    np.testing.assert_allclose(np.abs(f).sum(axis=(1, 2)), gold["features_abs_sum"], rtol=1e-5)


def test_encoder_batch32_matches_oracle_and_is_batch_invariant(eng, dev, synth_ckpt):
    from oracle.swin import encoder_forward
    img = W.synthetic_images(32)
    f = eng.encode(img.to(dev)).cpu()
    ref = encoder_forward(img[[0, 17, 31]], synth_ckpt["encoder"])
    assert (f[[0, 17, 31]] - ref).abs().max().item() < 5e-5
    f1 = eng.encode(img[17:18].contiguous().to(dev)).cpu()
    assert torch.equal(f1[0], f[17]), "per-image results must not depend on the batch they were computed in"


def _check_greedy(out, gold, max_len):
    lens = out["lengths"].cpu().numpy()
    toks = out["tokens"].cpu().numpy()
    lp = out["token_logp"].cpu().numpy()
    hid = out["hidden"].cpu().numpy()
    assert lens.tolist() == gold["lens"].tolist()
    for b in range(len(lens)):
        n = int(lens[b])
        assert toks[b, :n].tolist() == gold["ids"][b, :n].tolist(), f"row {b}"
        assert np.abs(lp[b, :n] - gold["token_logp"][b, :n]).max() < 1e-3
        assert np.abs(hid[b, :min(8, n)] - gold["hidden_head"][b, :min(8, n)]).max() < 1e-3
        assert np.abs(hid[b, :n].astype(np.float64).sum(0) - gold["hidden_sum"][b]).max() < 2e-2


def test_greedy_decode_vs_reference_golden_with_compaction(golden_dir, eng, dev):
    """Rows finish at different steps: exercises the batch-row positional-encoding quirk under compaction."""
    gold = np.load(os.path.join(golden_dir, "decoder_greedy.npz"))
    feats = W.hash_normal("decoder_greedy_features", (6, 144, 1024), 0.5).to(dev)
    out = eng.decode_greedy(feats, trace_logits=True)
    _check_greedy(out, gold, 480)
    lg = out["logits"].cpu().numpy()
    for s in range(4):
        assert np.abs(lg[s] - gold[f"logits_step{s}"]).max() < 1e-3      # north_star: 1e-3 on logits


def test_greedy_decode_max_length_finish(golden_dir, eng, dev):
    gold = np.load(os.path.join(golden_dir, "decoder_short.npz"))
    feats = W.hash_normal("decoder_short_features", (3, 144, 1024), 0.5).to(dev)
    _check_greedy(eng.decode_greedy(feats, max_len=24), gold, 24)


def test_chunk_ids_emulate_separate_reference_batches(eng, dev, synth_ckpt):
    """Two reference batches decoded in ONE engine call must equal two separate oracle runs (PE rows restart per chunk)."""
    from oracle.decoder import greedy_decode
    feats = W.hash_normal("chunk_features", (5, 144, 1024), 0.5)
    out = eng.decode_greedy(feats.to(dev), chunk_id=torch.tensor([0, 0, 0, 1, 1]), max_len=160)
    ref_a = greedy_decode(feats[:3], synth_ckpt["decoder"], max_len=160)
    ref_b = greedy_decode(feats[3:], synth_ckpt["decoder"], max_len=160)
    lens = out["lengths"].cpu().tolist()
    toks = out["tokens"].cpu().numpy()
    for b, ref in enumerate(ref_a.tokens + ref_b.tokens):
        assert toks[b, :lens[b]].tolist() == ref, f"row {b}"


def test_decode_batch32_natural_lengths_vs_oracle(eng, dev, synth_ckpt):
    """Full-size workload: B=32 encoder features -> greedy decode until EOS/480 -> tokens exact vs the oracle
    given the SAME features; then size-independent properties of the outputs."""
    from oracle.decoder import greedy_decode
    feats = eng.encode(W.synthetic_images(32).to(dev))
    out = eng.decode_greedy(feats)
    ref = greedy_decode(feats.cpu(), synth_ckpt["decoder"])
    lens = out["lengths"].cpu().tolist()
    toks = out["tokens"].cpu().numpy()
    assert lens == [len(t) for t in ref.tokens]
    for b in range(32):
        assert toks[b, :lens[b]].tolist() == ref.tokens[b], f"row {b}"
        seq = toks[b, :lens[b]]
        assert (seq[:-1] != 2).all()                                   # EOS only as the last token
        assert lens[b] == 480 or seq[-1] == 2
        prev_x = (seq[:-1] >= 101) & (seq[:-1] < 165)
        assert ((seq[1:][prev_x] >= 165)).all()                        # grammar: x-bin is followed by a y-bin
        prev_y = seq[:-1] >= 165
        assert (seq[1:][prev_y] < 101).all()                           # grammar: no coordinate after a y-bin
    assert len(set(lens)) > 8, "workload should contain many different lengths"


def test_beam1_equals_greedy(eng, dev):
    """beam_size=1, n_best=1 must reproduce greedy search (natural lengths, compaction of finished images)."""
    feats = W.hash_normal("decoder_features", (6, 144, 1024), 0.5).to(dev)
    g = eng.decode_greedy(feats)
    b = eng.decode_beam(feats, beam=1, n_best=1)
    assert b["lengths"][:, 0].cpu().tolist() == g["lengths"].cpu().tolist()
    for i, n in enumerate(g["lengths"].cpu().tolist()):
        assert torch.equal(b["tokens"][i, 0, :n], g["tokens"][i, :n]), f"row {i}"
        assert (b["hidden"][i, 0, :n] - g["hidden"][i, :n]).abs().max().item() < 1e-5
        lp = g["token_logp"][i, :n].double().sum().item() / (n + 1)      # reference normalisation: tokens + <sos>
        assert abs(b["scores"][i, 0].item() - lp) < 1e-3 * max(1.0, abs(lp))


def _engine_with_tick(synth_ckpt, tile, tile_ff=4, fused_max=128, slots=128, branch_rows=0, branch_max=4, xcd=0, mid_max=0):
    """An engine whose greedy tick runs fused on the given row tiles (tile 0: the 8-launches-per-layer tick of decoder.hip),
    ticks of more than branch_rows rows as up to branch_max parallel branches of rows (0: one chain); ticks of more than
    fused_max and up to mid_max rows take the mid form (dec_ma + dec_mb in place of dec_fa)."""
    from molnextr_amd.engine import Engine
    keys = {"MNX_DEC_TILE": str(tile), "MNX_DEC_TILE_FF": str(tile_ff), "MNX_DEC_FUSED_MAX": str(fused_max),
            "MNX_DEC_MID_MAX": str(mid_max),
            "MNX_DEC_BRANCH_ROWS": str(branch_rows), "MNX_DEC_BRANCH_MAX": str(branch_max), "MNX_DEC_XCD": str(xcd)}
    old = {k: os.environ.get(k) for k in keys}
    os.environ.update(keys)
    try:
        return Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=32, dec_slots=slots, dtype="fp16x3")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_fused_tick_is_independent_of_the_row_tiles_and_matches_the_unfused_tick(eng, dev, synth_ckpt):
    """dec_fused.hip defines its arithmetic per element (fixed chains, fixed trees), not per workgroup shape: 2 or 4 rows per
    attention workgroup, 4 / 8 / 16 rows per feed-forward workgroup must give BIT-identical hidden states and log-probs; the
    8-launches-per-layer tick of decoder.hip sums the same products in another order: same tokens, hidden within 1e-4. Rows
    finish at different steps (compaction, PE quirk) and run up to position 479 (second key per thread, value loop tail)."""
    feats = eng.encode(W.synthetic_images(32).to(dev))
    outs = {}
    for tiles in ((4, 4), (2, 8), (4, 16), (0, 4), (2, 4, "xcd"), (4, 8, "xcd"), (4, 4, "mid")):
        mid = len(tiles) > 2 and tiles[2] == "mid"      # the 32-row tick on the mid form: fused only up to 16 rows
        e = _engine_with_tick(synth_ckpt, tiles[0], tiles[1], xcd=int(len(tiles) > 2 and not mid), fused_max=16 if mid else 128,
                              mid_max=4096 if mid else 0)
        try:
            a = e.decode_greedy(feats)
            b = e.decode_greedy(feats[:5].contiguous(), max_len=480, stop_on_eos=False)
            outs[tiles] = tuple({k: v.cpu() for k, v in o.items() if v is not None} for o in (a, b))
        finally:
            e.close()
    # "xcd": row tiles pinned to XCDs (another grid order); "mid": dec_fa cut into its linear half on 16-row tiles and its
    # attention half (dec_ma / dec_mb), every key of the row read from the cache — the same chains on the same numbers
    for tiles in ((2, 8), (4, 16), (2, 4, "xcd"), (4, 8, "xcd"), (4, 4, "mid")):
        for x, y in zip(outs[(4, 4)], outs[tiles]):
            assert torch.equal(x["lengths"], y["lengths"]), f"tiles {tiles}"
            for i, n in enumerate(x["lengths"].tolist()):
                assert torch.equal(x["tokens"][i, :n], y["tokens"][i, :n]), f"tiles {tiles} row {i}"
                assert torch.equal(x["hidden"][i, :n], y["hidden"][i, :n]), f"tiles {tiles} row {i}: hidden states differ bitwise"
                assert torch.equal(x["token_logp"][i, :n], y["token_logp"][i, :n]), f"tiles {tiles} row {i}"
    for x, y in zip(outs[(4, 4)], outs[(0, 4)]):
        assert torch.equal(x["lengths"], y["lengths"])
        for i, n in enumerate(x["lengths"].tolist()):
            assert torch.equal(x["tokens"][i, :n], y["tokens"][i, :n]), f"row {i}"
            assert (x["hidden"][i, :n] - y["hidden"][i, :n]).abs().max().item() < 1e-4
            assert (x["token_logp"][i, :n] - y["token_logp"][i, :n]).abs().max().item() < 1e-4


def test_fused_and_unfused_ticks_mix_in_one_job(eng, dev, synth_ckpt):
    """mnx_predict picks the tick by capacity: with MNX_DEC_FUSED_MAX=64 a 160-image job starts on the decoder.hip tick
    (capacity 192) and drains on the fused one. Tokens / atoms / bonds must equal the all-fused and the all-unfused job."""
    imgs = W.synthetic_images(160, first_index=500).to(dev)
    res = []
    # (tile, fused_max, branch_rows, branch_max): the last three vary how a tick is cut into parallel branches of rows —
    # one chain, two branches of 96 rows (fused), five of 32, two of 96 on the 8-launch form — which must not change anything
    # the last two: the job starts on the MID form (capacity 192 > 64 / 128) and drains on the fused one; mid form throughout
    for tile, fmax, brows, bmax, xcd, mmax in ((4, 64, 128, 4, 0, 0), (4, 4096, 0, 1, 0, 0), (0, 0, 0, 1, 0, 0), (2, 128, 128, 4, 0, 0),
                                               (4, 128, 32, 8, 0, 0), (0, 0, 96, 2, 0, 0), (-1, 4096, 0, 1, 1, 0),
                                               (-1, 64, 0, 1, 0, 4096), (-1, 128, 0, 1, 0, 4096), (4, 16, 0, 1, 0, 4096)):
        e = _engine_with_tick(synth_ckpt, tile, 4, fmax, slots=256, branch_rows=brows, branch_max=bmax, xcd=xcd, mid_max=mmax)
        try:
            res.append({k: v.cpu() for k, v in e.predict(imgs, ref_batch=32).items()})
        finally:
            e.close()
    for r in res[1:]:
        _same_predictions(res[0], r)


def test_persistent_encoder_grids_on_fewer_cus_give_identical_predictions(eng, dev, synth_ckpt):
    """MNX_ENC_CUS=n launches the encoder's persistent kernels (gemm256x3_kernel, window_attn_pipe_kernel) on n (2 n)
    workgroups so that 256 - n CUs stay free for the decode stream. Which workgroup computes a tile, and how a
    layer's rows are split between the 256x256 and the 128x128 kernel, changes with n; the results must not (every GEMM kernel
    adds an element's terms in one order). The setting is process-wide: restored to 256 at the end."""
    from molnextr_amd.engine import Engine
    imgs = W.synthetic_images(96, first_index=900).to(dev)
    want = {k: v.cpu() for k, v in eng.predict(imgs, ref_batch=32).items()}
    fwant = eng.encode(imgs[:32].contiguous()).cpu()
    old = os.environ.get("MNX_ENC_CUS")
    try:
        for n in (224, 192):
            os.environ["MNX_ENC_CUS"] = str(n)
            e = Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=96, dec_slots=128, dtype="fp16x3")
            try:
                assert torch.equal(e.encode(imgs[:32].contiguous()).cpu(), fwant), f"features differ bitwise at {n} workgroups"
                _same_predictions(want, {k: v.cpu() for k, v in e.predict(imgs, ref_batch=32).items()})
            finally:
                e.close()
        with pytest.raises(Exception, match="MNX_ENC_CUS"):
            os.environ["MNX_ENC_CUS"] = "32"
            Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=32, dec_slots=64, dtype="fp16x3")
    finally:
        os.environ["MNX_ENC_CUS"] = "256"
        Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=32, dec_slots=64, dtype="fp16x3").close()
        if old is None:
            os.environ.pop("MNX_ENC_CUS", None)
        else:
            os.environ["MNX_ENC_CUS"] = old


@pytest.mark.parametrize("B,beam,n_best,max_len", [(4, 3, 2, 160), (3, 5, 5, 96), (2, 8, 1, 64), (5, 2, 2, 480),
                                                   (8, 5, 2, 128),      # 40 rows: two 32-row tiles
                                                   (32, 5, 1, 96),      # BASELINE config 5: beam 5 x batch 32 = 5 tiles
                                                   (32, 8, 3, 48)])     # the capacity corner: 256 rows, 8 tiles
def test_beam_search_vs_oracle(eng, dev, synth_ckpt, B, beam, n_best, max_len):
    """Beam search through the C ABI against oracle/beam.py (whose strategy is pinned on the reference's BeamSearch
    class): hypotheses, their order, scores and the decoder outputs along each hypothesis."""
    from oracle.beam import beam_decode
    feats = W.hash_normal(f"beam_features_{B}_{beam}", (B, 144, 1024), 0.5)
    ref = beam_decode(feats, synth_ckpt["decoder"], beam=beam, n_best=n_best, max_len=max_len)
    out = eng.decode_beam(feats.to(dev), beam=beam, n_best=n_best, max_len=max_len)
    lens = out["lengths"].cpu().numpy()
    toks = out["tokens"].cpu().numpy()
    sc = out["scores"].cpu().numpy()
    for i in range(B):
        assert len(ref.tokens[i]) == n_best
        for r in range(n_best):
            assert toks[i, r, :lens[i, r]].tolist() == ref.tokens[i][r], f"image {i} rank {r}"
            assert abs(sc[i, r] - ref.scores[i][r]) < 1e-4 * max(1.0, abs(ref.scores[i][r]))
            n = lens[i, r]
            assert (out["hidden"][i, r, :n].cpu() - ref.hidden[i][r]).abs().max().item() < 1e-3
    assert all(sc[i, r] >= sc[i, r + 1] for i in range(B) for r in range(n_best - 1))
    # (no 'beam >= greedy' property: an image leaves the search when its TOP beam finishes, so a short hypothesis
    #  can end the search below the score greedy reaches later — observed on these inputs)


def test_beam_capacity_errors(eng, dev):
    from molnextr_amd.engine import MnxError
    feats = W.hash_normal("beam_err", (2, 144, 1024), 0.5).to(dev)
    with pytest.raises(MnxError):
        eng.decode_beam(feats, beam=9, n_best=1, max_len=16)
    with pytest.raises(MnxError):
        eng.decode_beam(feats, beam=2, n_best=3, max_len=16)


def test_facade_beam_matches_engine_beam_and_bond_head(eng, dev, synth_ckpt):
    """decode_batch(beam_size=3): best hypothesis detokenised, bond head on ITS decoder outputs (checked against the
    oracle's beam + bond head)."""
    from molnextr_amd.model import decode_batch
    from oracle.beam import beam_decode
    from oracle.edges import predict_edges
    from molnextr_amd.tokenizer import get_tokenizer
    tok = get_tokenizer()["chartok_coords"]
    feats = W.hash_normal("beam_facade", (3, 144, 1024), 0.5)
    preds = decode_batch(eng, feats.to(dev), beam_size=3, n_best=2, ref_batch_size=3)
    ref = beam_decode(feats, synth_ckpt["decoder"], beam=3, n_best=2)
    for i, p in enumerate(preds):
        d = tok.sequence_to_smiles(ref.tokens[i][0])
        assert p["chartok_coords"]["smiles"] == d["smiles"] and p["chartok_coords"]["indices"] == d["indices"]
        assert len(p["beam_scores"]) == 2
        if d["indices"]:
            e, _ = predict_edges(ref.hidden[i][0], d["indices"], synth_ckpt["decoder"])
            assert p["edges"] == np.asarray(e).astype(int).tolist()


def test_fixed_length_decode_is_deterministic(eng, dev):
    feats = W.hash_normal("det_features", (4, 144, 1024), 0.5).to(dev)
    a = eng.decode_greedy(feats, max_len=64, stop_on_eos=False)
    b = eng.decode_greedy(feats, max_len=64, stop_on_eos=False)
    assert a["lengths"].cpu().tolist() == [64] * 4
    assert torch.equal(a["tokens"], b["tokens"]) and torch.equal(a["hidden"], b["hidden"])


@pytest.mark.parametrize("name,T", [("a", 40), ("b", 90), ("c", 12), ("d", 20)])
def test_bond_head_vs_reference_golden(golden_dir, eng, dev, name, T):
    gold = np.load(os.path.join(golden_dir, "edges.npz"))
    hidden = torch.zeros(1, 480, 256)
    hidden[0, :T] = W.hash_normal(f"edges_hidden_{name}", (T, 256), 1.0)
    idx = gold[f"{name}_idx"]
    k = len(idx)
    ai = torch.zeros(1, 160, dtype=torch.int32)
    ai[0, :k] = torch.from_numpy(idx)
    e, s = eng.edges(hidden.to(dev), ai.to(dev), torch.tensor([k], dtype=torch.int32), want_scores=True)
    assert np.array_equal(e.cpu().numpy()[0, :k, :k], gold[f"{name}_edges"])
    assert np.abs(s.cpu().numpy()[0, :k, :k] - gold[f"{name}_scores"]).max() < 1e-5


def test_bond_head_ragged_batch_and_empty(eng, dev, synth_ckpt):
    from oracle.edges import predict_edges
    hidden = W.hash_normal("edges_ragged", (4, 480, 256), 1.0)
    ks = [0, 1, 57, 159]
    ai = torch.zeros(4, 160, dtype=torch.int32)
    for b, k in enumerate(ks):
        order = torch.argsort(W.hash_normal(f"ragged_idx{b}", (480,), 1.0))      # distinct positions: no exact ties
        ai[b, :k] = torch.sort(order[:k])[0].int()
    e, _ = eng.edges(hidden.to(dev), ai.to(dev), torch.tensor(ks, dtype=torch.int32))
    e = e.cpu().numpy()
    for b, k in enumerate(ks):
        ref, _ = predict_edges(hidden[b], ai[b, :k].tolist(), synth_ckpt["decoder"])
        assert np.array_equal(e[b, :k, :k], ref), f"sample {b} (k={k})"
        up = np.triu(e[b, :k, :k], 1)
        lo = np.tril(e[b, :k, :k], -1).T
        swap = np.array([0, 1, 2, 3, 4, 6, 5])
        assert np.array_equal(np.triu(swap[lo], 1), up)                # symmetric, wedge <-> dash mirrored


def test_end_to_end_predictions_vs_reference_golden(golden_dir, eng, dev):
    """Decoder.decode end to end (tokens -> host detokenise -> bond head) against the reference's own output."""
    from molnextr_amd.model import decode_batch
    with open(os.path.join(golden_dir, "predict_e2e.json")) as f:
        gold = json.load(f)["preds"]
    feats = W.hash_normal("e2e_features", (4, 144, 1024), 0.5).to(dev)
    preds = decode_batch(eng, feats)
    for p, g in zip(preds, gold):
        c = p["chartok_coords"]
        assert c["smiles"] == g["smiles"] and c["symbols"] == g["symbols"] and c["indices"] == g["indices"]
        assert c["coords"] == g["coords"]
        assert p["edges"] == g["edges"]


def test_predict_pipeline_equals_per_batch_path(eng, dev):
    """mnx_predict (continuous batching: rows of several reference batches resident at once, on-device atom scan)
    must give, image by image, exactly what the per-batch API + host tokenizer give."""
    from molnextr_amd.tokenizer import get_tokenizer
    tok = get_tokenizer()["chartok_coords"]
    imgs = W.synthetic_images(80).to(dev)              # reference batches of 32: 32 + 32 + 16
    out = eng.predict(imgs, ref_batch=32)
    lens = out["lengths"].cpu().numpy()
    toks = out["tokens"].cpu().numpy()
    na = out["n_atoms"].cpu().numpy()
    ai = out["atom_idx"].cpu().numpy()
    ed = out["edges"].cpu().numpy()
    for first in (0, 32, 64):
        x = imgs[first:first + 32].contiguous()
        feats = eng.encode(x)
        ref = eng.decode_greedy(feats)
        rl = ref["lengths"].cpu().numpy()
        rt = ref["tokens"].cpu().numpy()
        n = x.shape[0]
        idx = torch.zeros(n, 160, dtype=torch.int32)
        cnt = torch.zeros(n, dtype=torch.int32)
        for b in range(n):
            i = first + b
            assert lens[i] == rl[b] and toks[i, :lens[i]].tolist() == rt[b, :rl[b]].tolist(), f"image {i}"
            d = tok.sequence_to_smiles(rt[b, :rl[b]].tolist())
            assert na[i] == len(d["indices"]) and ai[i, :na[i]].tolist() == d["indices"], f"image {i}: atom positions"
            cnt[b] = len(d["indices"])
            idx[b, :cnt[b]] = torch.tensor(d["indices"], dtype=torch.int32)
        e, _ = eng.edges(ref["hidden"], idx.to(dev), cnt)
        e = e.cpu().numpy()
        for b in range(n):
            k = int(cnt[b])
            assert np.array_equal(ed[first + b, :k, :k], e[b, :k, :k]), f"image {first + b}: bonds"


def _same_predictions(a, b):
    a = {k: v.cpu().numpy() for k, v in a.items()}
    b = {k: v.cpu().numpy() for k, v in b.items()}
    assert np.array_equal(a["lengths"], b["lengths"]) and np.array_equal(a["n_atoms"], b["n_atoms"])
    for i, (n, k) in enumerate(zip(a["lengths"], a["n_atoms"])):
        assert np.array_equal(a["tokens"][i, :n], b["tokens"][i, :n]), f"image {i}: tokens"
        assert np.array_equal(a["atom_idx"][i, :k], b["atom_idx"][i, :k]), f"image {i}: atom positions"
        assert np.array_equal(a["edges"][i, :k, :k], b["edges"][i, :k, :k]), f"image {i}: bonds"


def test_grouped_encode_gives_identical_predictions(eng, dev, synth_ckpt):
    """An engine with max_batch=64 encodes several reference batches per encoder call (bigger GEMMs); the encoder is
    batch-invariant, so tokens / atoms / bonds must equal the max_batch=32 engine's bit for bit (ragged last group)."""
    from molnextr_amd.engine import Engine
    imgs = W.synthetic_images(88, first_index=200).to(dev)       # reference batches of 16: 5 full + one of 8
    a = eng.predict(imgs, ref_batch=16)
    big = Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=64, dtype="fp16x3")
    try:
        b = big.predict(imgs, ref_batch=16)
        c = big.predict(imgs[:40].contiguous(), ref_batch=32)   # group of one full + one ragged reference batch
    finally:
        big.close()
    _same_predictions(a, b)
    _same_predictions(c, eng.predict(imgs[:40].contiguous(), ref_batch=32))


def test_predict_results_do_not_depend_on_neighbouring_work(eng, dev):
    """Continuous batching keeps sequences of many reference batches in one set of slots; a batch's results must not
    depend on what else is resident (slot tiles, tick sizes, admission order): the last batch of a 96-image job equals
    the same 32 images decoded alone."""
    imgs = W.synthetic_images(96, first_index=300).to(dev)
    whole = eng.predict(imgs, ref_batch=32)
    alone = eng.predict(imgs[64:].contiguous(), ref_batch=32)
    _same_predictions({k: v[64:] for k, v in whole.items()}, alone)


def test_decode_is_bit_reproducible_under_concurrent_encoder_load(eng, dev):
    """One batch decoded alone, then again while another stream keeps the chip busy with the encoder's store-heavy
    GEMMs: tokens AND hidden states must be bit-identical. (Round 2 found a build whose per-row decode kernels were
    reproducible alone but not next to the encoder: loads through re-used 64-bit VGPR address pairs returned wrong
    data under memory back-pressure — DESIGN.md §6. The continuous-batching path always decodes next to the encoder.)"""
    imgs = W.synthetic_images(32).to(dev)
    f = eng.encode(imgs)
    torch.cuda.synchronize()
    ref = eng.decode_greedy(f, max_len=64, stop_on_eos=False)
    side = torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(side):
            for _ in range(2):
                eng.encode(imgs)
        r = eng.decode_greedy(f, max_len=64, stop_on_eos=False)
        torch.cuda.synchronize()
        assert torch.equal(r["tokens"], ref["tokens"]) and torch.equal(r["hidden"], ref["hidden"])


def test_device_atom_scan_vs_reference_golden_and_fuzz(golden_dir, eng, dev):
    """The on-device restatement of sequence_to_smiles' 'indices': reference golden cases + a fuzz against the host
    tokenizer (itself pinned by the same golden cases)."""
    import random
    from molnextr_amd.tokenizer import get_tokenizer
    tok = get_tokenizer()["chartok_coords"]
    with open(os.path.join(golden_dir, "tokenizer.json")) as f:
        seqs = [(c["ids"], c["out"]["indices"]) for c in json.load(f)["cases"] if len(c["ids"]) <= 480]
    rng = random.Random(7)
    s = tok.stoi
    for _ in range(400):
        seq = []
        n = rng.randint(0, 200)
        while len(seq) < n:
            r = rng.random()
            if r < 0.55:
                sym = rng.choice(["C", "N", "O", "Cl", "Br", "[", "c", "*", "<unk>", "B", "Cr"])
                if sym == "[":
                    seq += [s["["], s[rng.choice("CNOH@+-23")], s[rng.choice("CNOH@+-23]")], s["]"]]
                elif sym == "<unk>":
                    seq.append(3)
                else:
                    seq += [s[c] for c in sym]
                if rng.random() < 0.9:
                    seq += [101 + rng.randint(0, 63), 165 + rng.randint(0, 63)]
            else:
                seq.append(rng.randint(0, 228))
        seqs.append((seq, tok.sequence_to_smiles(seq)["indices"]))
    T = 480
    tokens = torch.zeros(len(seqs), T, dtype=torch.int32)
    lengths = torch.zeros(len(seqs), dtype=torch.int32)
    for i, (ids, _) in enumerate(seqs):
        tokens[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        lengths[i] = len(ids)
    idx, cnt = eng.atom_scan(tokens.to(dev), lengths.to(dev))
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for i, (ids, want) in enumerate(seqs):
        assert cnt[i] == len(want) and idx[i, :cnt[i]].tolist() == want, (i, ids[:30])


def test_facade_reference_batch_size_semantics(eng, dev, synth_ckpt):
    """decode_batch(ref_batch_size=16) must equal the reference run with batch_size=16 (its default in
    predict_images): rows 0-15 and 16-31 are separate positional-encoding numberings."""
    from molnextr_amd.model import decode_batch
    from oracle.decoder import greedy_decode
    feats = W.hash_normal("facade_features", (24, 144, 1024), 0.5)
    preds = decode_batch(eng, feats.to(dev), ref_batch_size=16, compute_confidence=True, max_len=200)
    ref = greedy_decode(feats[:16], synth_ckpt["decoder"], max_len=200).tokens + \
        greedy_decode(feats[16:], synth_ckpt["decoder"], max_len=200).tokens
    from molnextr_amd.tokenizer import get_tokenizer
    tok = get_tokenizer()["chartok_coords"]
    for p, r in zip(preds, ref):
        d = tok.sequence_to_smiles(r)
        assert p["chartok_coords"]["smiles"] == d["smiles"] and p["chartok_coords"]["indices"] == d["indices"]
        k = len(d["symbols"])
        assert len(p["edges"]) == k and len(p["chartok_coords"]["atom_scores"]) == k
        assert 0.0 <= p["overall_score"] <= 1.0
        assert all(0.0 < s <= 1.0 for s in p["chartok_coords"]["atom_scores"])


def test_confidence_outputs_vs_reference_golden(golden_dir, eng, dev):
    """Decoder.decode(compute_confidence=True): atom / edge / overall scores against the reference's own output."""
    from molnextr_amd.model import decode_batch
    with open(os.path.join(golden_dir, "predict_e2e_conf.json")) as f:
        gold = json.load(f)["preds"]
    feats = W.hash_normal("conf_features", (3, 144, 1024), 0.5).to(dev)
    preds = decode_batch(eng, feats, compute_confidence=True)
    for p, g in zip(preds, gold):
        c = p["chartok_coords"]
        assert c["smiles"] == g["smiles"] and c["indices"] == g["indices"]
        np.testing.assert_allclose(c["atom_scores"], g["atom_scores"], rtol=2e-4)
        es = np.array(p["edge_scores"])
        np.testing.assert_allclose(es.sum(axis=1), g["edge_score_row_sums"], rtol=1e-5)
        np.testing.assert_allclose(np.log(es).sum(), g["edge_score_log_sum"], rtol=1e-4)
        if g["edge_scores"] is not None:
            np.testing.assert_allclose(es, np.array(g["edge_scores"]), atol=1e-5)
        assert abs(p["overall_score"] - g["overall_score"]) <= 1e-6 + 1e-3 * abs(g["overall_score"])


def test_pipeline_facade_equals_per_batch_facade(eng, dev):
    """predict_pipeline (mnx_predict) and encode + decode_batch must give identical per-image dicts, with the
    reference's default batch_size=16 as the numbering unit."""
    from molnextr_amd.model import decode_batch, predict_pipeline
    imgs = W.synthetic_images(40, first_index=200).to(dev)
    a = predict_pipeline(eng, imgs, ref_batch_size=16)
    b = []
    for i in range(0, 40, 32):
        b += decode_batch(eng, eng.encode(imgs[i:i + 32].contiguous()), ref_batch_size=16)
    assert len(a) == len(b) == 40
    for x, y in zip(a, b):
        assert x == y


def test_predict_beam_pipeline_equals_per_batch_beam(eng, dev):
    """mnx_predict_beam (encoder running ahead on its own stream, beam search batch by batch, device atom scan and bond
    head on the winner) must give exactly what encode + mnx_decode_beam + host atom positions + mnx_edges give per
    reference batch — 40 images as reference batches of 16 (two full, one ragged), beam 5."""
    from molnextr_amd.model import decode_batch, predict_pipeline
    imgs = W.synthetic_images(40, first_index=300).to(dev)
    a = predict_pipeline(eng, imgs, ref_batch_size=16, max_len=96, beam_size=5)
    b = []
    for i in range(0, 40, 16):
        b += decode_batch(eng, eng.encode(imgs[i:i + 16].contiguous()), ref_batch_size=16, max_len=96, beam_size=5)
    assert len(a) == len(b) == 40
    for x, y in zip(a, b):
        assert x["chartok_coords"] == y["chartok_coords"] and x["edges"] == y["edges"]
        assert abs(x["beam_scores"][0] - y["beam_scores"][0]) < 1e-6


def test_beam_search_of_several_reference_batches_in_one_step_sequence_equals_separate_searches(dev, synth_ckpt):
    """mnx_predict_beam searches up to eight reference batches of an encoder launch group in ONE step sequence (BASELINE config
    5: 8 x 32 images x 5 hypotheses = 1280 rows per step instead of 160). Images are independent but for the positional-encoding
    row, which beam_begin_kernel numbers inside each image's own reference batch while rows of all batches share
    the step — so the hypotheses must be EXACTLY those of separate searches: 240 images as reference batches of 32 (seven
    full, one ragged) through the pipeline with groups of 8, 3 and 1, against batch-by-batch mnx_decode_beam + host atom
    positions + mnx_edges; beam 5 and beam 8 (8 x 32 x 5 = 1280 rows of capacity: at beam 8 the group count is capped by the slots)."""
    from molnextr_amd.engine import Engine
    from molnextr_amd.model import decode_batch, predict_pipeline
    eng = Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=256, dec_slots=1280, dtype="fp16x3")
    try:
        imgs = W.synthetic_images(240, first_index=500).to(dev)
        for beam, max_len in ((5, 128), (8, 64)):
            ref = []
            for i in range(0, 240, 32):
                ref += decode_batch(eng, eng.encode(imgs[i:i + 32].contiguous()), ref_batch_size=32, max_len=max_len, beam_size=beam)
            for groups in ("8", "3", "1"):
                os.environ["MNX_BEAM_GROUPS"] = groups
                try:
                    got = predict_pipeline(eng, imgs, ref_batch_size=32, max_len=max_len, beam_size=beam)
                finally:
                    os.environ.pop("MNX_BEAM_GROUPS", None)
                assert len(got) == len(ref) == 240
                for i, (x, y) in enumerate(zip(got, ref)):
                    assert x["chartok_coords"] == y["chartok_coords"] and x["edges"] == y["edges"], (beam, groups, i)
                    assert abs(x["beam_scores"][0] - y["beam_scores"][0]) < 1e-6, (beam, groups, i)
    finally:
        eng.close()


def test_facade_uploads_the_next_group_while_the_engine_works(dev):
    """predict_images on host pages in several engine calls: group g+1 is uploaded (pinned staging) and transformed by
    mnx_preprocess on a side stream / helper thread while mnx_predict runs group g. The results must be those of the
    single-call path image by image (groups are whole reference batches, so the numbering does not change)."""
    from molnextr_amd.model import molnextr
    m = molnextr("synthetic", dev, max_batch=8)
    pages = [W.synthetic_page(c) for c in range(11)] + [np.full((90, 130, 3), 255, np.uint8)]
    for i, p in enumerate(pages[-1:]):
        p[30:60, 20 + i:100] = 0
    one = m.predict_images(pages, return_atoms_bonds=True, batch_size=4)
    m.group_images = 4                                   # 3 engine calls, two of them overlapped with an upload
    many = m.predict_images(pages, return_atoms_bonds=True, batch_size=4)
    assert len(one) == len(many) == 12 and one == many
    m.engine.close()


def test_public_api_predict_images_synthetic(dev):
    """molnextr('synthetic').predict_images: reference output dict keys; no RDKit here -> SMILES fields None."""
    from molnextr_amd.model import molnextr, BOND_TYPES
    m = molnextr("synthetic", dev, max_batch=4)
    img = np.full((120, 200, 3), 255, np.uint8)
    img[40:80, 60:140] = 0
    img[55:65, 20:180] = 30
    out = m.predict_images([img, img[:, ::-1].copy()], return_atoms_bonds=True, return_confidence=True)
    assert len(out) == 2
    for o in out:
        assert set(o) == {"predicted_smiles", "predicted_molfile", "atom_sets", "bond_sets"}
        for a in o["atom_sets"]:
            assert set(a) == {"atom_number", "atom_symbol", "coords", "confidence"}
            assert 0.0 <= a["coords"][0] <= 1.0 and 0.0 <= a["coords"][1] <= 1.0
        for b in o["bond_sets"]:
            assert b["bond_type"] in BOND_TYPES[1:] and b["endpoints"][0] < b["endpoints"][1]
    m.engine.close()


def test_device_preprocess_is_bit_identical_to_host_restatement(eng, dev):
    """mnx_preprocess (bounding box + virtual white border + fixed-point bilinear + gray + normalise) against
    molnextr_amd/preprocess.py on ragged pages: blank, 1x1, ink on the borders, upscaling, strong decimation."""
    from molnextr_amd.preprocess import transform_image
    rng = np.random.default_rng(7)
    pages = []
    for (h, w) in [(470, 923), (64, 64), (1, 1), (37, 911), (1500, 2000), (384, 384), (300, 17)]:
        img = np.full((h, w, 3), 255, np.uint8)
        for _ in range(12):       # random coloured strokes, some touching the page border
            y, x = rng.integers(0, h), rng.integers(0, w)
            hh, ww = rng.integers(1, max(2, h // 3)), rng.integers(1, max(2, w // 3))
            img[y:y + hh, x:x + ww] = rng.integers(0, 256, size=3, dtype=np.uint8)
        pages.append(img)
    pages.append(np.full((50, 70, 3), 255, np.uint8))                       # blank page: no crop, border only
    pages.append(rng.integers(0, 256, size=(200, 333, 3), dtype=np.uint8))  # noise: exercises every weight pair
    pages.append(rng.integers(0, 256, size=(90, 120), dtype=np.uint8))      # single-channel input
    edge = np.full((40, 40, 3), 255, np.uint8); edge[0, 0] = 0; edge[-1, -1] = 254
    pages.append(edge)
    out = eng.preprocess(pages).cpu().numpy()
    sq = eng.preprocess(pages, pad_to_square=True).cpu().numpy()
    for i, p in enumerate(pages):
        ref = transform_image(p)
        assert np.array_equal(out[i], ref), f"page {i} {p.shape}: {np.abs(out[i] - ref).max()}"
        ref = transform_image(p, square=True)
        assert np.array_equal(sq[i], ref), f"page {i} {p.shape} (PadToSquare): {np.abs(sq[i] - ref).max()}"


def test_device_crop_box_vs_reference_golden(golden_dir, eng, dev):
    """The device bounding-box kernel against CropWhite.update_params of the reference's own data_aug.py
    (tests/golden/crop_pad.json, ragged pages regenerated from their recipe), and the whole device transform with and
    without PadToSquare against the host restatement whose crop / pad stages are pinned on the same fixture."""
    from molnextr_amd.preprocess import transform_image
    with open(os.path.join(golden_dir, "crop_pad.json")) as f:
        cases = json.load(f)["cases"]
    pages = [W.synthetic_page(c["case"]) for c in cases]
    out, crops = eng.preprocess(pages, return_crops=True)
    crops = crops.cpu().numpy()
    sq = eng.preprocess(pages, pad_to_square=True).cpu().numpy()
    out = out.cpu().numpy()
    for c, page, crop in zip(cases, pages, crops):
        assert crop.tolist() == c["crop"], (c["case"], crop.tolist(), c["crop"])
        assert np.array_equal(out[c["case"]], transform_image(page))
        assert np.array_equal(sq[c["case"]], transform_image(page, square=True))


def test_reference_format_checkpoint_loads_through_public_api(dev, tmp_path):
    """Rehearsal for the day molnextr_best.pth is supplied (BASELINE configs 1 and 4): the synthetic checkpoint written
    in the reference's training format (DDP 'module.' prefixes, optimizer / scheduler / scaler state, saved args —
    main.py:389-398) loads through molnextr(model_path=...) and gives the same predictions as the in-memory one; the
    safetensors conversion of it too."""
    from molnextr_amd import checkpoint as C
    from molnextr_amd.model import molnextr
    ck = W.synthetic_checkpoint(0)
    pth = {"encoder": {"module." + k: v for k, v in ck["encoder"].items()},
           "decoder": {"module." + k: v for k, v in ck["decoder"].items()},
           "optimizer": {"state": {0: {"exp_avg": torch.zeros(4)}}, "param_groups": []}, "scheduler": {"last_epoch": 3},
           "scaler": {"scale": 65536.0}, "global_step": 1234, "epoch": 7,
           "args": {"formats": ["chartok_coords", "edges"], "input_size": 384, "coord_bins": 64, "sep_xy": True,
                    "encoder": "swin_base", "decoder": "transformer"}}
    src, dst = str(tmp_path / "molnextr_best.pth"), str(tmp_path / "molnextr_best.safetensors")
    torch.save(pth, src)
    C.convert(src, dst)
    pages = [W.synthetic_page(0), W.synthetic_page(5)]
    outs = []
    for path in ("synthetic", src, dst):
        m = molnextr(path, dev, max_batch=4)
        outs.append(m.predict_images(pages, return_atoms_bonds=True))
        m.engine.close()
    assert outs[0] == outs[1] == outs[2]
    assert outs[0][0]["atom_sets"] is not None
    with pytest.raises(ValueError, match="checkpoint path is required"):
        molnextr(None, dev)


def test_facade_falls_back_to_the_bf16_split_mode_when_fp16_overflows(dev, tmp_path):
    """A checkpoint the reference runs without complaint but whose MLP hidden state leaves the fp16 range (fc1 of one block
    x 3e5, its fc2 x 1 / 3e5: activations ~1e6): the default fp16x3 engine (and an fp16x3m one) reports MNX_ERR_RANGE; the facade must warn,
    rebuild in bf16x3 (fp32 exponent range) and return what the fp32 oracle returns — not raise. A later call on a sane
    input must not inherit the flag (it is per call)."""
    from molnextr_amd.engine import DEFAULT_DTYPE, Engine, MnxError, MNX_ERR_RANGE
    from molnextr_amd.model import molnextr, predict_pipeline
    from molnextr_amd.tokenizer import get_tokenizer
    from oracle.decoder import greedy_decode
    from oracle.swin import encoder_forward
    ck = W.synthetic_checkpoint(0)
    p = "transformer.layers.2.blocks.7.mlp."
    ck["encoder"][p + "fc1.weight"] *= 3e5
    ck["encoder"][p + "fc1.bias"] *= 3e5
    ck["encoder"][p + "fc2.weight"] /= 3e5
    src = str(tmp_path / "big_activations.pth")
    torch.save({"encoder": ck["encoder"], "decoder": ck["decoder"], "args": ck["args"]}, src)
    imgs = W.synthetic_images(3, first_index=40)
    # the engine alone reports the range error, with its code
    e = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=4, dec_slots=64, dtype="fp16x3")
    try:
        with pytest.raises(MnxError) as ei:
            e.predict(imgs.to(dev), ref_batch=3)
        assert ei.value.code == MNX_ERR_RANGE
    finally:
        e.close()
    m = molnextr(src, dev, max_batch=4)
    try:
        assert m.engine.dtype == DEFAULT_DTYPE == "fp16x3"
        with pytest.warns(RuntimeWarning, match="bf16x3"):
            preds = m._with_fallback(lambda eng: predict_pipeline(eng, imgs.to(dev), m.tokenizer, ref_batch_size=3))
        assert m.engine.dtype == "bf16x3"
        ref = greedy_decode(encoder_forward(imgs, ck["encoder"]), ck["decoder"])
        tok = get_tokenizer()["chartok_coords"]
        for pr, ids in zip(preds, ref.tokens):
            d = tok.sequence_to_smiles(ids)
            assert pr["chartok_coords"]["smiles"] == d["smiles"] and pr["chartok_coords"]["indices"] == d["indices"]
        # the rebuilt engine serves the next call without another warning
        again = m._with_fallback(lambda eng: predict_pipeline(eng, imgs.to(dev), m.tokenizer, ref_batch_size=3))
        assert [q["chartok_coords"]["smiles"] for q in again] == [q["chartok_coords"]["smiles"] for q in preds]
    finally:
        m.engine.close()


def test_range_flag_is_per_call(eng, dev):
    """mnx_encode on an input that overflows sets the device flag; a caller that never polls mnx_encoder_status must not
    make the NEXT mnx_predict (valid images) fail with MNX_ERR_RANGE."""
    bad = torch.full((1, 3, 384, 384), float("inf"), device=dev)
    eng.encode(bad)
    out = eng.predict(W.synthetic_images(2).to(dev), ref_batch=2)     # would raise MnxError(-6) if the flag were sticky
    assert int(out["lengths"].min()) >= 1
    assert not eng.encoder_nonfinite()


def test_eval_harness_reproduces_reference_batches(eng, dev, tmp_path):
    """molnextr_amd.evaluate.run_inference on image files: DistributedSampler-style shard (world 1 and a simulated
    rank of world 2), per-rank batches of batch_size*2 as reference batches, records -> prediction dicts; must equal the
    per-batch engine path on the same batches."""
    from PIL import Image
    from molnextr_amd import evaluate as E
    from molnextr_amd.preprocess import load_image_rgb
    from molnextr_amd.tokenizer import get_tokenizer
    tok = get_tokenizer()["chartok_coords"]
    rng = np.random.default_rng(3)
    paths = []
    for i in range(11):
        page = np.full((120 + 10 * i, 200, 3), 255, np.uint8)
        for _ in range(10):
            y, x = rng.integers(0, 100), rng.integers(0, 180)
            page[y:y + rng.integers(1, 20), x:x + rng.integers(1, 20)] = 0
        paths.append(str(tmp_path / f"p{i}.png"))
        Image.fromarray(page).save(paths[-1])
    load = lambda i: load_image_rgb(paths[i])
    preds = E.run_inference(eng, load, len(paths), batch_size=2)
    assert sorted(preds) == list(range(11))
    for b in E.reference_batches(E.sampler_indices(11, 0, 1), batch_size=2):     # batches of 4, 4, 3
        feats = eng.encode(eng.preprocess([load(i) for i in b]))
        out = eng.decode_greedy(feats)
        for r, i in enumerate(b):
            n = int(out["lengths"][r])
            assert preds[i]["chartok_coords"]["smiles"] == tok.sequence_to_smiles(out["tokens"][r, :n].cpu().tolist())["smiles"]
    table = E.predictions_table([f"p{i}" for i in range(11)], preds)
    assert len(table["SMILES"]) == 11 and table["edges"][0].startswith("[")


def test_capacity_and_argument_errors(eng, dev):
    from molnextr_amd.engine import MnxError
    with pytest.raises(MnxError, match="32"):
        eng.decode_greedy(torch.zeros(33, 144, 1024, device=dev))
    with pytest.raises(MnxError):
        eng.decode_greedy(torch.zeros(2, 144, 1024, device=dev), max_len=481)


def test_strict_weight_validation(synth_ckpt, dev):
    from molnextr_amd.engine import Engine
    bad = dict(synth_ckpt["encoder"])
    bad.pop("transformer.layers.2.blocks.7.attn.qkv.weight")
    with pytest.raises(ValueError, match="missing transformer.layers.2.blocks.7.attn.qkv.weight"):
        Engine(bad, synth_ckpt["decoder"])
