"""Shape constants of the path. Test infrastructure (see oracle/__init__.py)."""
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class SwinConfig:
    """MolNexTR/models/transformers.py:547-551 (`swin_base`), :423-428 (defaults)."""
    img_size: int = 384
    patch: int = 4
    embed_dim: int = 128
    depths: Tuple[int, ...] = (2, 2, 18, 2)
    heads: Tuple[int, ...] = (4, 8, 16, 32)
    window: int = 12

    @property
    def num_features(self):
        return self.embed_dim * 2 ** (len(self.depths) - 1)

    @property
    def grid(self):
        return self.img_size // self.patch


@dataclass(frozen=True)
class DecoderConfig:
    """MolNexTR/model.py:60-76 (decoder flags), MolNexTR/utils.py:25 (max_len 480),
    MolNexTR/tokenization.py:172-178 (token id ranges)."""
    layers: int = 6
    d_model: int = 256
    heads: int = 8
    d_ff: int = 1024
    vocab: int = 229
    sym_offset: int = 101   # first x-bin id  (tokenizer.offset)
    bins: int = 64          # x bins [101,165), y bins [165,229)
    max_len: int = 480
    enc_dim: int = 1024
    pad_id: int = 0
    sos_id: int = 1
    eos_id: int = 2
    edge_classes: int = 7


SWIN_B_384 = SwinConfig()
DECODER_DEFAULT = DecoderConfig()
