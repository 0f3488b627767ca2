"""CPU checks of arithmetic that the HIP kernels hard-code (no GPU, no compute through the library): the constants are
read from the kernel sources and evaluated here in float32 exactly as the kernel evaluates them."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gelu_fast4_source():
    src = open(os.path.join(ROOT, "molnextr_amd", "csrc", "common.h")).read()
    body = src[src.index("f32x4 gelu_fast4(f32x4 x) {"):]
    return body[:body.index("\n}\n")]


def test_polynomial_gelu_of_the_gemm_epilogues_vs_exact_erf_gelu():
    """gelu_fast4 (common.h): max(x, -5) * (0.5 + xc * Q(u)), xc = clamp(x, +-5), u = 0.08 xc^2 - 1, Q by Horner in
    float32. The encoder's fc1 epilogue stores the result as bf16 / fp16 (half-ulp >= 2.4e-4 relative): the approximation
    has to stay orders below that — here |error| <= 3e-6 absolute over [-1e4, 1e4] (the negative tail must not grow with
    |x|) and <= 2e-3 relative where |gelu| >= 1e-3."""
    body = _gelu_fast4_source()
    lead = re.search(r"q = u \* ([-0-9.e+]+)f \+ ([-0-9.e+]+)f;", body)
    rest = re.findall(r"q = q \* u \+ ([-0-9.e+]+)f;", body)
    coef = [float(lead.group(1)), float(lead.group(2))] + [float(c) for c in rest]
    assert len(coef) == 13                                     # degree 12 in u
    assert "* 0.08f - 1.0f" in body and "-5.0f, 5.0f" in body and "xo * (xc * q + 0.5f)" in body
    assert body.count("fmaxf(x[") == 4
    f = np.float32
    x = np.concatenate([np.linspace(-30, 30, 1200001), np.linspace(-1e4, 1e4, 200001),
                        np.array([0.0, -0.0, 5.0, -5.0, 1e-30, -1e-30, -1e30])]).astype(f)
    xc = np.clip(x, f(-5), f(5))
    u = (xc * xc * f(0.08) - f(1)).astype(f)
    q = (u * f(coef[0]) + f(coef[1])).astype(f)
    for c in coef[2:]:
        q = (q * u + f(c)).astype(f)
    got = (np.maximum(x, f(-5)) * (xc * q + f(0.5)).astype(f)).astype(np.float64)
    x64 = x.astype(np.float64)
    want = x64 * 0.5 * (1.0 + erf(x64 / np.sqrt(2.0)))
    err = np.abs(got - want)
    assert err.max() < 3e-6
    big = np.abs(want) >= 1e-3
    assert (err[big] / np.abs(want[big])).max() < 2.5e-3
    assert got[np.argmax(x == 0)] == 0.0


def test_degree16_polynomial_gelu_is_as_accurate_as_the_reference_formula():
    """gelu_poly16 (common.h; lab option MNX_GELU_POLY of the split-mode epilogues, measured: no faster on the power-limited
    GEMMs): x Phi(x) with Phi = 0.5 + xc Q(u), xc = clamp(x, +-5.5), u = 2 xc^2 / 5.5^2 - 1, Q of degree 16, every
    step one FMA. Its header claims an error of <= 1.3e-7 max(1, |x|) against the exact function — as good as the formula the
    reference evaluates, 0.5 x (1 + erf(x / sqrt 2)) in fp32 with a correctly rounded erf (1.1e-7 max(1, |x|))."""
    src = open(os.path.join(ROOT, "molnextr_amd", "csrc", "common.h")).read()
    body = src[src.index("f32x4 gelu_poly16(f32x4 x) {"):]
    body = body[:body.index("\n}\n")]
    lead = re.search(r"q = u \* ([-0-9.e+]+)f \+ ([-0-9.e+]+)f;", body)
    rest = re.findall(r"q = q \* u \+ ([-0-9.e+]+)f;", body)
    coef = [float(lead.group(1)), float(lead.group(2))] + [float(c) for c in rest]
    assert len(coef) == 17 and "-5.5f, 5.5f" in body and "xo * (xc * q + 0.5f)" in body
    scale = float(re.search(r"xc \* xc \* ([-0-9.e+]+)f - 1.0f", body).group(1))
    assert abs(scale - 2.0 / 5.5 ** 2) < 1e-9
    f = np.float32

    def fma(a, b, c):        # a * b is exact in float64 for float32 inputs: one rounding, as the hardware FMA
        return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(f)
    x = np.concatenate([np.linspace(-8, 8, 1600001), np.random.default_rng(0).normal(size=400000) * 1.5]).astype(f)
    xc = np.clip(x, f(-5.5), f(5.5))
    u = fma(xc * xc, np.full_like(x, f(scale)), f(-1.0))
    q = fma(u, np.full_like(x, f(coef[0])), f(coef[1]))
    for c in coef[2:]:
        q = fma(q, u, f(c))
    got = (np.maximum(x, f(-5.5)) * fma(xc, q, f(0.5))).astype(np.float64)
    x64 = x.astype(np.float64)
    want = 0.5 * x64 * (1.0 + erf(x64 / np.sqrt(2.0)))
    scale_x = np.maximum(1.0, np.abs(x64))
    err = np.abs(got - want) / scale_x
    e32 = erf((x * f(0.7071067811865476)).astype(np.float64)).astype(f)
    ref = ((f(0.5) * x) * (f(1) + e32)).astype(np.float64)
    err_ref = np.abs(ref - want) / scale_x
    assert err.max() < 1.4e-7 and err_ref.max() < 1.2e-7, (err.max(), err_ref.max())
    assert np.sqrt((err ** 2).mean()) < 3.0e-8


def test_fma_mix_split_is_the_two_rounding_split():
    """common.h split16x4_mix derives the lo plane as ONE rounding of fma(hi, -1, v) to fp16 (v_fma_mix{lo,hi}_f16) where
    split16x4 rounds twice: t = fl32(v - hi), lo = RN16(t). They agree bit for bit iff v - hi is exactly representable in fp32
    for every fp32 v and hi = RN16(v) — checked here on 4M values covering the normal range, the fp16 subnormal range (where hi
    is a multiple of 2^-24), the overflow edge and the probabilities x 2^10 the window attention actually splits (the GPU check
    of the same statement is tools/attn_lab's word compare: profiles/r06_attn_lab_diet.txt)."""
    rng = np.random.default_rng(5)
    v = np.concatenate([
        rng.standard_normal(1 << 20).astype(np.float32) * np.float32(3.0),
        (rng.random(1 << 20).astype(np.float32) * np.float32(1024.0)),                                   # P x 2^10
        np.exp2(rng.uniform(-30, -10, 1 << 20)).astype(np.float32) * rng.choice(np.float32([-1, 1]), 1 << 20),   # subnormal hi / lo
        np.exp2(rng.uniform(10, 15.99, 1 << 20)).astype(np.float32),                                     # up to the fp16 maximum
    ])
    hi = v.astype(np.float16)
    ok = np.isfinite(hi.astype(np.float32))
    d64 = v.astype(np.float64) - hi.astype(np.float64)                 # exact
    d32 = v - hi.astype(np.float32)                                     # what __fsub_rn gives
    assert np.array_equal(d64[ok], d32[ok].astype(np.float64)), "v - RN16(v) must be exact in fp32"
    lo_two = d32.astype(np.float16)                                     # split16x4: second rounding of the fp32 difference
    lo_one = d64.astype(np.float16)                                     # split16x4_mix: one rounding of the exact fma
    assert np.array_equal(lo_two[ok].view(np.uint16), lo_one[ok].view(np.uint16))
    # and the pair carries v to 2^-22 |v| (or 2^-25 absolute in the subnormal range)
    err = np.abs(hi[ok].astype(np.float64) + lo_one[ok].astype(np.float64) - v[ok].astype(np.float64))
    assert np.all(err <= np.maximum(np.abs(v[ok]).astype(np.float64) * 2.0 ** -22, 2.0 ** -25))


def test_gemm256_vmcnt_bookkeeping_constants():
    """The counted waits of gemm256.hip are derived from how many vector-memory operations a wave issues per tile;
    the constants the derivation uses must match the code that issues them."""
    src = open(os.path.join(ROOT, "molnextr_amd", "csrc", "gemm256.hip")).read()
    assert "constexpr int P_STORES = 16, P_BIAS = 4;" in src
    epi = src[src.index("auto epilogue = [&]() {"):]
    epi = epi[:epi.index("    };\n")]
    # 8 slabs x 2 sixteen-byte stores per lane and plane (split modes store a second, lo plane: PST = 2 * P_STORES),
    # nothing else that touches vector memory
    assert "constexpr int PST = SPLIT ? 2 * P_STORES : P_STORES;" in src
    assert "for (int mt = 0; mt < 8; ++mt)" in epi and epi.count("for (int i = 0; i < 2; ++i)") == 2
    assert epi.count("= o8;") == 2 and epi.count("if (SPLIT) {") == 2 and "resid" not in epi
    bias = src[src.index("auto load_bias = [&](int n0) {"):]
    bias = bias[:bias.index("    };\n")]
    assert "for (int nt = 0; nt < 4; ++nt)" in bias and bias.count("global_load_dwordx4") == 1
    assert src.count("wait_vm<8 + PST + P_BIAS>()") == 2 and src.count("wait_vm<8 + P_BIAS>()") == 1


def test_gemm256x3_isa_has_no_scratch_and_only_its_own_m0_writes(tmp_path):
    """gemm256x3_kernel (gemm256.hip) counts vector-memory operations by hand (s_waitcnt vmcnt(N)) and issues its LDS-DMA
    through inline asm that writes M0, a register the compiler does not track around asm. Both only hold if the generated
    code (1) has no scratch (spill) traffic — scratch loads / stores are vector-memory operations too —, (2) touches M0
    nowhere but in the `s_mov_b32 m0` in front of each of those DMA instructions, and (3) contains no wait the compiler
    added on its own: exactly the vmcnt(0) of the drain branches and the counts derived in the kernel, for the six-phase
    (three terms) and the four-phase (two terms) loop alike."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "molnextr_amd", "csrc", "gemm256.hip")
    out = tmp_path / "gemm256.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN3mnx12_GLOBAL__N_116gemm256x3_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    # bf16: three terms x {bias, bias + GELU, bias + residual fp32, bias fp32}; fp16: {three, two} terms x those four + the GELU
    # epilogue that keeps the hi output plane only (its consumer runs on two terms)
    assert len(kernels) == 4 + 2 * 5
    seen = set()
    for name, body in kernels:
        t, epi, terms, lo_out = re.search(r"kernelI(DF16_|DF16b)Li(\d)ELi(\d)ELb([01])E", name).groups()
        epi, terms, lo_out = int(epi), int(terms), lo_out == "1"
        seen.add((t, epi, terms, lo_out))
        assert "scratch_" not in body, name
        n_dma = len(re.findall(r"global_load_lds_dword", body))
        assert len(re.findall(r"\bm0\b", body)) == n_dma == len(re.findall(r"s_mov_b32 m0,", body)), name
        waits = sorted(int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
        out16 = epi in (0, 1)
        # vector-memory operations of a tile's epilogue (+ 1 bias DMA) that are younger than the fills a first-K-tile wait covers
        pst = (32 if lo_out else 16) if out16 else (64 if epi == 2 else 32)
        if terms == 3:      # phase waits P1 P2 P4 P5 P6 (P3 waits for nothing): younger fills 10 10 12 12 2, prologue 8
            normal, first, prologue = [10, 10, 12, 12, 2], [10 + pst + 1, 10 + pst + 1, 12 + pst + 1, 12 + 1, 2], 8
        else:               # phase waits P1 P2 P4: younger fills 6 6 6, prologue 6
            normal, first, prologue = [6, 6, 6], [6 + pst + 1, 6 + pst + 1, 6], 6
        first = [min(w, 63) for w in first]
        if out16:
            # one copy of the K-tile body: prologue DMAs (bias + every slot K-tile 0 needs) + 1 bias in P1 + the K-tile's fills; no
            # compiler-made wait; 2 barriers per phase + 2 in the prologue + the closing one
            assert n_dma == (32 if terms == 3 else 26), (name, n_dma)
            assert waits == sorted([0] * len(normal) + normal + first + [prologue]), (name, waits)
            assert len(re.findall(r"s_barrier", body)) == (15 if terms == 3 else 11), name
        else:
            # fp32 epilogues: residual loads and stores are inline asm on ONE scalar base per array + a 32-bit lane offset, retired
            # by counted waits — four slabs of residual in flight, 12 16 20 24 24 20 16 12 younger operations allowed when slab
            # 0..7 is consumed. The compiler sees no vector-memory operation in the epilogue, so it adds no wait of its own (it
            # may duplicate the K-tile body and the two copies of the epilogue)
            resid = epi == 2
            expect = {}
            for w in normal + first + [prologue]:
                expect[w] = expect.get(w, 0) + 1
            for w, n in expect.items():
                if w in (12, 16, 20, 24) and resid:
                    continue                                          # shared with the epilogue's own counted waits, checked below
                assert waits.count(w) >= n and waits.count(w) % n == 0, (name, w, waits.count(w))
            copies = waits.count(24) // 2 if resid else 0
            if resid:
                assert copies >= 2 and waits.count(16) == 2 * copies and waits.count(20) == 2 * copies, (name, waits)
                assert waits.count(12) >= 2 * copies + (2 if terms == 3 else 0), (name, waits)     # + the two phase waits of 12
            allowed = set(normal + first + [prologue, 0]) | ({12, 16, 20, 24} if resid else set())
            assert set(waits) <= allowed, (name, sorted(set(waits) - allowed))     # anything else would be a compiler-made wait
            ld = re.findall(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\]", body)
            st = re.findall(r"global_store_dwordx4 v\d+, v\[\d+:\d+\], s\[\d+:\d+\]", body)
            n_ld = len(re.findall(r"global_load_dwordx4", body)), len(re.findall(r"global_store_dwordx4", body))
            assert (len(ld), len(st)) == n_ld and len(st) % 32 == 0 and len(st) >= 64, (name, n_ld)
            assert len(ld) == (len(st) if resid else 0), (name, len(ld), len(st))
            # the hazard the compiler cannot see through asm: a VALU / LDS-read / load result written into the DATA registers of
            # a 16-byte store within two wait states of it (gfx940+). The first build had one wait state: one wrong word in 512
            lines = [l.strip() for l in body.split("\n")]
            lines = [l for l in lines if l and not l.startswith(";") and not l.startswith(".")]
            def regs(tok):
                m = re.match(r"v\[(\d+):(\d+)\]", tok)
                if m:
                    return set(range(int(m.group(1)), int(m.group(2)) + 1))
                m = re.match(r"v(\d+)$", tok)
                return {int(m.group(1))} if m else set()
            for i, l in enumerate(lines):
                if not l.startswith("global_store_dwordx4"):
                    continue
                data = regs(l.split()[2].strip(","))
                assert len(data) == 4, l
                ws, k = 0, i + 1
                while ws < 2 and k < len(lines):
                    nxt = lines[k]
                    if nxt.startswith("s_nop"):
                        ws += int(nxt.split()[1]) + 1
                    else:
                        if nxt.startswith(("v_", "ds_read", "global_load_dwordx4")):
                            assert not (regs(nxt.split()[1].strip(",")) & data), (name, l, nxt)
                        ws += 1
                    k += 1
            # the asm loads / stores take the tile's base from ONE SGPR pair per array, produced once in front of the
            # epilogue's `s_nop 4` (gfx950 needs wait states between a SALU write of an SGPR and a vector-memory instruction that
            # uses it as its address base; the hazard recognizer does not look into asm): no scalar instruction between the first
            # and the last of these accesses may write a base register, and nothing may copy a load's destination before the
            # counted wait has retired it (the compiler believes the asm's output is valid immediately)
            mem = [i for i, l in enumerate(lines) if re.match(r"global_(load|store)_dwordx4 .*s\[\d+:\d+\]", l)]
            # every copy of the epilogue starts at its `s_nop 4` (the bases are final there) and ends at its 32nd store
            starts = [i for i, l in enumerate(lines) if l == "s_nop 4"]
            assert len(starts) == len(st) // 32, (name, len(starts), len(st))
            for s0 in starts:
                k, n_st, bases = s0 + 1, 0, set()
                while n_st < 32:
                    l = lines[k]
                    if re.match(r"global_(load|store)_dwordx4 .*s\[\d+:\d+\]", l):
                        m = re.search(r"s\[(\d+):(\d+)\]", l)
                        bases |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                        n_st += l.startswith("global_store")
                    k += 1
                for l in lines[s0 + 1:k]:
                    if l.startswith("s_") and not l.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_cbranch", "s_branch", "s_endpgm")):
                        dst = l.split()[1].strip(",")
                        m = re.match(r"s\[(\d+):(\d+)\]", dst)
                        d = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else ({int(dst[1:])} if re.match(r"s\d+$", dst) else set())
                        assert not (d & bases), (name, "a scalar instruction rewrites an asm address base inside the epilogue", l)
            for i in mem:
                if not lines[i].startswith("global_load_dwordx4"):
                    continue
                dest = regs(lines[i].split()[1].strip(","))
                k = i + 1
                while k < len(lines) and not lines[k].startswith("s_waitcnt vmcnt"):
                    if lines[k].startswith(("v_mov", "v_accvgpr_write", "scratch_")):
                        srcs = set().union(*[regs(t.strip(",")) for t in lines[k].split()[2:]]) if len(lines[k].split()) > 2 else set()
                        assert not (srcs & dest), (name, "a load destination is copied before its wait", lines[i], lines[k])
                    k += 1
    assert seen == ({("DF16b", e, 3, True) for e in range(4)} | {("DF16_", e, t, True) for e in range(4) for t in (2, 3)}
                    | {("DF16_", 1, t, False) for t in (2, 3)})


def test_window_attn_pipe_isa_fits_two_workgroups_per_cu_and_never_drains_the_fetch_queue(tmp_path):
    """window_attn_pipe_kernel (encoder.hip) keeps the next item's K / V LDS-DMA and q loads in flight while the current
    item is multiplied. That only works if the generated code (1) fits 80 VGPRs with NO scratch (two 9-wave workgroups
    per CU put 6 waves on SIMD 0; a scratch reload is a vector-memory operation whose wait drains the queue), (2) waits
    for vector memory exactly once per item — the hand-written vmcnt(0) at the top, plus the compiler's copy of it in
    front (it waits for the prefetched registers that pass through the asm) — and once more for the prologue, (3) issues
    4 DMA per item (prologue + loop = 8) through `s_mov_b32 m0` and touches M0 nowhere else, (4) reads V through the
    transposing LDS read (5 key blocks x 2 channel halves x 2 planes x 2 halves = 40, minus the 4 of the absent second
    half of block 4, which re-reads the first half: 36)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "molnextr_amd", "csrc", "encoder.hip")
    out = tmp_path / "encoder.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN3mnx23window_attn_pipe_kernel\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M)
    assert len(kernels) == 4                                   # {fp16, bf16} planes x {windows without, with} shift mask
    for name, body in kernels:
        meta = body
        assert "scratch_" not in body, name
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)) <= 80, name
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)) == 0, name
        n_dma = len(re.findall(r"global_load_lds_dwordx4", body))
        assert n_dma == 8 and len(re.findall(r"\bm0\b", body)) == n_dma == len(re.findall(r"s_mov_b32 m0,", body)), name
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)
        assert waits and all(w == "0" for w in waits) and len(waits) <= 3, (name, waits)
        loop = body[body.index("s_barrier"):]
        assert len(re.findall(r"s_waitcnt vmcnt", loop)) == 0, name          # nothing between the barrier and the loop's back edge ...
        assert len(re.findall(r"s_barrier", body)) == 1, name                 # ... and one barrier per item
        assert len(re.findall(r"ds_read_b64_tr_b16", body)) == 36, name
        # round 6: the unmasked fp16 kernel splits P and the context through v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16 (common.h
        # split16x4_mix: 10 blocks of four probabilities + 2 of the context), every block behind its own wait state (a VALU
        # read of a transcendental's result; the compiler does not guard inline asm); the other three keep the compiler's form
        mix = len(re.findall(r"v_fma_mixlo_f16", body)), len(re.findall(r"v_fma_mixhi_f16", body))
        # ... and the unmasked kernels take their byte offsets from LDS tables: 4 integer multiplies left in the whole item
        # (the masked ones, whose windows wrap around the image, re-derive them: 16)
        loop_body = body[body.index("Inner Loop Header"):]
        n_mul = len(re.findall(r"v_mul_lo_u32|v_mul_hi_u32|v_mad_u64_u32", loop_body))
        assert n_mul <= (6 if "Lb0E" in name else 20), (name, n_mul)
        if "IDF16_Lb0E" in name:
            assert mix == (24, 24), (name, mix)
            assert len(re.findall(r"s_nop 0\n\s*v_cvt_pk_f16_f32 v\d+, v\d+, v\d+\n\s*v_cvt_pk_f16_f32", body)) == 12, name
        else:
            assert mix == (0, 0), (name, mix)


def test_patch_embed_isa_requests_its_staging_loads_together_and_has_a_branch_free_tap_loop(tmp_path):
    """patch_embed_kernel<CPT> (encoder.hip) was 0.18 of its HBM bound while the channel count was a run-time argument: every
    weight quad sat in a branch of its own (LDS read, wait, 4 FMAs), and the staging loops waited for each global load before
    storing it to LDS. What the rewrite rests on, in the ISA of the C = 128 instantiation: (1) no scratch in any
    instantiation; (2) all of a thread's staging loads (6 weight quads, 5 pixel quads, bias) are issued before the first wait
    for vector memory; (3) between the second barrier (tiles staged) and the first store there are the tap loop's back edge
    and the `px < G` guard, no other branch, no wait for vector memory other than the one for the hoisted gamma / beta, and
    one line's worth of packed FMAs (4 taps x 16 channels x 3 patches / 2 = 96)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "molnextr_amd", "csrc", "encoder.hip")
    out = tmp_path / "encoder.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    kernels = dict(re.findall(r"^(_ZN3mnx18patch_embed_kernelILi\d+E\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M))
    assert len(kernels) == 4                                   # C = 32, 64, 96, 128
    for name, body in kernels.items():
        assert "scratch_" not in body, name
    body = next(b for n, b in kernels.items() if "ILi16E" in n)
    first_wait = re.search(r"s_waitcnt vmcnt", body).start()
    assert len(re.findall(r"global_load_dwordx4", body[:first_wait])) >= 6 + 5 + 4
    barriers = [m.start() for m in re.finditer(r"s_barrier", body)]
    assert len(barriers) == 2
    taps = body[barriers[1]:body.index("global_store_dwordx4")]
    assert len(re.findall(r"s_cbranch", taps)) == 2, re.findall(r"s_cbranch\w*", taps)
    # (round 6: a workgroup walks two patch rows, so the chunk body is a real loop and the compiler keeps one more counted wait
    #  behind the barrier for loads it hoisted across the iteration, next to the one for gamma / beta)
    assert len(re.findall(r"s_waitcnt vmcnt", taps)) <= 2
    assert len(re.findall(r"v_pk_fma_f32", taps)) >= 96
    assert len(re.findall(r"ds_read_b128", taps)) >= 4 * 4 + 3          # a line: 4 taps x 4 weight quads, 3 pixel quads


def test_dec_attn_isa_has_its_value_rows_in_flight_before_it_waits_for_a_key(tmp_path):
    """dec_attn_kernel (decoder.hip): the first five value block-loads are requested ahead of the keys, so that the kernel's
    first wait for vector memory has 5 value quads (8 + 4 + 4 bytes each of the 24-bit block format, kvq.h) and the thread's
    key row (6 x 16 bytes + its scale) outstanding (before: the values were fetched after the softmax, two more dependent
    round trips), the second key row of rows past 256 keys behind ONE uniform branch in front of that wait too; the row is
    decoded by one SDWA / byte convert per stored element; sgemm_tn_kernel: the next K slice's two loads sit between the two
    barriers of the current slice (under its FMAs), not in front of the LDS stores."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "molnextr_amd", "csrc", "decoder.hip")
    out = tmp_path / "decoder.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    kernels = dict(re.findall(r"^(_ZN3mnx15dec_attn_kernelILb[01]E\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M))
    assert len(kernels) == 2
    for name, body in kernels.items():
        assert "scratch_" not in body, name
        first_wait = re.search(r"s_waitcnt vmcnt", body).start()
        if "ILb1E" in name:     # beam search: the slot holding each key comes from the ancestry table (five entries first)
            assert len(re.findall(r"global_load_dword ", body[:first_wait])) >= 5, name
        else:
            head = body[:first_wait]
            assert len(re.findall(r"global_load_dwordx2", head)) >= 5, name                  # value hi pieces
            assert len(re.findall(r"global_load_dwordx4", head)) >= 12, name                 # two key rows: 4 hi + 2 lo pieces each
            assert len(re.findall(r"global_load_dword ", head)) >= 5 * 2 + 2, name           # value lo pieces + scales, key scales
            assert len(re.findall(r"s_cbranch", head)) <= 2, name
            assert len(re.findall(r"v_cvt_f32_i32_sdwa", body)) >= 64 and len(re.findall(r"v_cvt_f32_ubyte", body)) >= 64, name
    sg = re.search(r"^_ZN3mnx15sgemm_tn_kernel\w+:[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M).group(1)
    loop = sg[sg.index("s_barrier"):]
    loop = loop[:loop.index("s_barrier", 10) + 9]              # first barrier .. second barrier of the K loop
    assert len(re.findall(r"global_load_dwordx4", loop)) == 2 and "ds_write" not in loop


def test_fused_decode_tick_isa_no_scratch_small_row_mfma_and_register_budget(tmp_path):
    """dec_fused.hip: the three kernels per decoder layer of the greedy tick. What its design rests on, checked in the ISA:
    (1) no scratch — the 1024-thread instantiations (4 rows per workgroup) have 128 registers per thread and every one of
    them holds a request in flight; a spill would put scratch round trips into an 8 us kernel; (2) the matrix work is the
    4-row instruction v_mfma_f32_4x4x1_16b_f32 (no 16-row tile on 2-4 row tiles); (3) the weights never pass through LDS
    (no LDS-DMA, and the LDS the kernels ask for is activations only: < 64 KB, so two workgroups fit a CU)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "molnextr_amd", "csrc", "dec_fused.hip")
    out = tmp_path / "dec_fused.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    # (a kernel has several s_endpgm: workgroups of dummy rows leave early; its text ends at .Lfunc_end)
    kernels = re.findall(r"^(_ZN3mnx13dec_f[abc]_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
    assert len(kernels) == 9          # fa {R = 2, 4} x {embedding, stream}, fb {2, 4}, fc {4, 8, 16}
    for name, body in kernels:
        assert "scratch_" not in body, name
        assert "v_mfma_f32_4x4x1_16b_f32" in body and "v_mfma_f32_16x16x4_f32" not in body, name
        assert "global_load_lds" not in body, name
    meta = dict(re.findall(r"\.name:\s+(_ZN3mnx13dec_f[abc]_kernel\w+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text))
    for name, vg in meta.items():
        if "ILi4E" in name and ("dec_fa" in name or "dec_fb" in name):
            assert int(vg) <= 128, (name, vg)             # 1024 threads per workgroup
