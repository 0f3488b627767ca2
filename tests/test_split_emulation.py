"""CPU: the operand-mode emulation of tools/study_split_terms.py (the tool behind DESIGN.md's split-term / FP8 table) on a
tiny Swin: its fp32 scheme IS the oracle, and the schemes order as the arithmetic says they must."""
import os
import sys

import numpy as np
import torch

from molnextr_amd import weights as W
from oracle.config import SwinConfig
from oracle.swin import encoder_forward

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import study_split_terms as ST  # noqa: E402

TINY_W = W.EncoderDims(img_size=96, patch=4, embed_dim=32, depths=(2, 2), heads=(1, 2), window=12)
TINY_O = SwinConfig(img_size=96, patch=4, embed_dim=32, depths=(2, 2), heads=(1, 2), window=12)


def test_emulated_schemes_on_a_tiny_swin():
    sd = W.synthetic_encoder_state(0, TINY_W)
    img = W.hash_normal("split_emulation_img", (2, 3, 96, 96), 1.0)
    ref = encoder_forward(img, sd, TINY_O)
    f32 = ST.encoder(img, sd, ST.Scheme("fp32"), TINY_O)
    assert (f32 - ref).abs().max().item() < 2e-5          # same ops; softmax written as exp / sum
    err = {}
    for name in ("fp16x3", "bf16x3", "fp16+fp8x2", "fp16x2", "fp16", "bf16"):
        err[name] = (ST.encoder(img, sd, ST.Scheme(name), TINY_O) - f32).abs().max().item()
    assert err["fp16x3"] < 2e-5
    assert err["fp16x3"] < err["bf16x3"] < err["fp16+fp8x2"] < err["fp16"] < err["bf16"], err
    assert err["bf16x3"] < err["fp16x2"], err              # dropping a correction term costs more than bf16 planes


def test_mx8_block_scaling():
    x = torch.tensor([[1.0, 0.5, 300.0, -448.0] + [0.0] * 28 + [1e-3] * 32])
    q = ST.mx8(x)
    assert q.shape == x.shape
    # block 0: amax 448 -> shared scale 2^0: 300 -> 288 or 320 (3 mantissa bits), 448 exact; block 1 has its own scale
    assert abs(q[0, 3].item() + 448.0) < 1e-6 and abs(q[0, 2].item() - 300.0) <= 20.0
    assert np.isclose(q[0, 40].item(), 1e-3, rtol=2 ** -4)


def test_two_term_tags_select_layers_by_class_and_stage():
    """--two tags ("cls" / "cls.sN"): a tagged product drops the activation's lo plane (ah.wh + ah.wl), the others keep three
    terms — the error of a two-term layer sits between fp16x3's and plain fp16's, and an untagged stage is untouched."""
    sd = W.synthetic_encoder_state(0, TINY_W)
    img = W.hash_normal("split_emulation_img", (2, 3, 96, 96), 1.0)
    f32 = ST.encoder(img, sd, ST.Scheme("fp32"), TINY_O)
    x3 = ST.encoder(img, sd, ST.Scheme("fp16x3"), TINY_O)
    two_all = ST.encoder(img, sd, ST.Scheme("fp16x3", ("fc1", "fc2")), TINY_O)
    two_s1 = ST.encoder(img, sd, ST.Scheme("fp16x3", ("fc1.s1", "fc2.s1")), TINY_O)
    none = ST.encoder(img, sd, ST.Scheme("fp16x3", ("fc1.s7",)), TINY_O)          # no such stage: nothing changes
    e = lambda t: (t - f32).abs().max().item()                                      # noqa: E731
    assert torch.equal(none, x3)
    assert e(x3) < e(two_s1) <= e(two_all) * 1.5 and e(two_all) < e(ST.encoder(img, sd, ST.Scheme("fp16"), TINY_O))
    assert e(two_all) > 5 * e(x3)


def test_block_fixed_point_rows_of_the_kv_cache():
    """round_block = what csrc/kvq.h stores per cached K / V row of 32 channels: integers of `bits` bits times one power-of-two
    scale per row; absolute error <= 2^-bits of the row's power-of-two ceiling, exact for rows that already are such integers,
    zeros stay zeros."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(50, 8, 32, generator=g) * torch.logspace(-6, 3, 50).reshape(50, 1, 1)
    for bits in (16, 20, 24):
        q = ST.round_block(x, bits)
        ceil2 = torch.exp2(torch.floor(torch.log2(x.abs().amax(-1, keepdim=True))) + 1.0)
        assert bool(((q - x).abs() <= ceil2 * 2.0 ** -bits * (1 + 1e-6)).all()), bits
        assert torch.equal(ST.round_block(q, bits), q), "idempotent"
    assert torch.equal(ST.round_block(torch.zeros(3, 32), 24), torch.zeros(3, 32))
    ints = torch.randint(-2 ** 22, 2 ** 22, (4, 32), generator=g).float() * 2.0 ** -30
    assert torch.equal(ST.round_block(ints, 24), ints)
