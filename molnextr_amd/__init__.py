"""molnextr_amd — MI355X-native engine behind the MolNexTR predict path.

Drop-in for the reference's `MolNexTR` package surface on that path: `get_predictions`,
`MolNexTRSingleton`, `molnextr` (reference MolNexTR/__init__.py, molnextr.py, model.py).
The Encoder/Decoder forward runs in hand-written HIP kernels (csrc/) behind a C-ABI library
(include/molnextr_hip.h) — there is no CPU fallback: if the library is missing, loading fails.
"""
__version__ = "0.1.0"

from .tokenizer import CharTokenizer, get_tokenizer  # noqa: F401
from .molnextr import MolNexTRSingleton, get_predictions  # noqa: F401

__all__ = ["get_predictions", "MolNexTRSingleton", "CharTokenizer", "get_tokenizer"]
