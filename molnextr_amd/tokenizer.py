"""Host-side token <-> chemistry glue for the 'chartok_coords' output format.

Mirrors the interface of the reference's CharTokenizer (reference MolNexTR/tokenization.py:330-515,
NodeTokenizer :112-199) for the inference direction only: ids -> raw SMILES, atom symbols, coordinates and
the decoder positions whose hidden states feed the bond head. Stays on the host by design (BASELINE
north_star); the grammar rule it defines (`get_output_mask`) is ALSO compiled into the HIP decode step
(csrc/decoder.hip) — `tests/test_tokenizer.py` checks both against the golden truth table.

Token id map (vocab/vocab_chars.json, 101 symbols): 0 <pad>, 1 <sos>, 2 <eos>, 3 <unk>, 4 <mask>,
5..100 characters, x-bins [101,165), y-bins [165,229); coordinate = (id - base) / 63.
"""
import json
import os

PAD_ID, SOS_ID, EOS_ID, UNK_ID, MASK_ID = 0, 1, 2, 3, 4
PAD, SOS, EOS, UNK, MASK = "<pad>", "<sos>", "<eos>", "<unk>", "<mask>"

_VOCAB_DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocab", "vocab_chars.json")


class CharTokenizer:
    """Character-level tokenizer with separate x / y coordinate bins (sep_xy=True)."""

    def __init__(self, input_size=64, path=None, sep_xy=True, continuous_coords=False, debug=False):
        with open(path or _VOCAB_DEFAULT) as f:
            self.stoi = json.load(f)
        self.itos = {i: s for s, i in self.stoi.items()}
        self.maxx = self.maxy = int(input_size)
        self.sep_xy = sep_xy
        self.continuous_coords = continuous_coords
        self.special_tokens = [PAD, SOS, EOS, UNK, MASK]
        self.debug = debug
        if not sep_xy or continuous_coords:
            raise NotImplementedError("the MolNexTR inference config uses sep_xy=True, discrete coordinates")

    # -- sizes / id classes (reference tokenization.py:125-170) --------------------------------
    def __len__(self):
        return self.offset + self.maxx + self.maxy

    @property
    def offset(self):
        return len(self.stoi)

    @property
    def output_constraint(self):
        return True

    def is_x(self, i):
        return self.offset <= i < self.offset + self.maxx

    def is_y(self, i):
        return self.offset + self.maxx <= i

    def is_symbol(self, i):
        return len(self.special_tokens) <= i < self.offset or i == UNK_ID

    @staticmethod
    def is_atom_token(token):
        return token.isalpha() or token.startswith("[") or token == "*" or token == UNK

    def is_atom(self, i):
        return self.is_symbol(i) and self.is_atom_token(self.itos[i])

    def id_to_x(self, i):
        return (i - self.offset) / (self.maxx - 1)

    def id_to_y(self, i):
        return (i - self.offset - self.maxx) / (self.maxy - 1)

    def x_to_id(self, x):
        return self.offset + round(x * (self.maxx - 1))

    def y_to_id(self, y):
        return self.offset + self.maxx + round(y * (self.maxy - 1))

    # -- decode-time grammar (reference tokenization.py:383-392) --------------------------------
    def get_output_mask(self, prev_id):
        """True = forbidden as the next token. After an x-bin only y-bins; after a y-bin no coordinate bins."""
        V, x0, y0 = len(self), self.offset, self.offset + self.maxx
        if self.is_x(prev_id):
            return [t < y0 for t in range(V)]
        if self.is_y(prev_id):
            return [t >= x0 for t in range(V)]
        return [False] * V

    # -- ids -> {smiles, symbols, coords, indices} (reference tokenization.py:464-515) ----------
    def _atom_span_end(self, seq, i):
        """End (exclusive) of the atom token starting at i: '[...]' up to the closing bracket or the first
        non-symbol id; 'Cl'/'Br' as two characters; otherwise one character."""
        first = self.itos[seq[i]]
        n = len(seq)
        if first == "[":
            j = i + 1
            while j < n and self.is_symbol(seq[j]):
                j += 1
                if self.itos[seq[j - 1]] == "]":
                    break
            return j
        if i + 1 < n and self.is_symbol(seq[i + 1]) and (first, self.itos[seq[i + 1]]) in (("C", "l"), ("B", "r")):
            return i + 2
        return i + 1

    def sequence_to_smiles(self, sequence):
        seq = list(sequence)
        n = len(seq)
        pieces, coords, symbols, indices = [], [], [], []
        i = 0
        while i < n:
            t = seq[i]
            if t == EOS_ID or t == PAD_ID:
                break
            if self.is_x(t) or self.is_y(t):
                i += 1
            elif not self.is_atom(t):
                pieces.append(self.itos[t])
                i += 1
            else:
                j = self._atom_span_end(seq, i)
                token = "".join(self.itos[seq[k]] for k in range(i, j))
                pieces.append(token)
                # an atom is kept only if "x y <next token>" all exist after it (the hidden state at the
                # position after y feeds the bond head)
                if j + 2 < n and self.is_x(seq[j]) and self.is_y(seq[j + 1]):
                    coords.append([self.id_to_x(seq[j]), self.id_to_y(seq[j + 1])])
                    symbols.append(token)
                    indices.append(j + 2)
                    i = j + 2
                else:
                    i = j
        return {"smiles": "".join(pieces), "symbols": symbols, "indices": indices, "coords": coords}


def get_tokenizer(args=None):
    """{'chartok_coords': CharTokenizer}; mirrors reference tokenization.py:518-543 for the inference formats."""
    bins = getattr(args, "coord_bins", 64) if args is not None else 64
    path = getattr(args, "vocab_file", None) if args is not None else None
    return {"chartok_coords": CharTokenizer(bins, path, sep_xy=True)}
