"""CPU restatement of the bond (edge) head. Test infrastructure only."""
import numpy as np
import torch
import torch.nn.functional as F

PE_ = "decoder.edges."


@torch.no_grad()
def edge_logits(hidden, atom_idx, sd):
    """GraphPredictor.forward, MolNexTR/components.py:365-377. hidden [T,256], atom_idx [k] -> logits [k,k,7].
    hh[i,j] = [h_i | h_j] -> Linear(512,256) -> GELU(erf) -> Linear(256,7)."""
    h = hidden[torch.as_tensor(atom_idx, dtype=torch.long)]
    k, d = h.shape
    hh = torch.cat([h[:, None, :].expand(k, k, d), h[None, :, :].expand(k, k, d)], dim=-1)
    z = F.gelu(F.linear(hh, sd[PE_ + "mlp.0.weight"], sd[PE_ + "mlp.0.bias"]))
    return F.linear(z, sd[PE_ + "mlp.2.weight"], sd[PE_ + "mlp.2.bias"])


def symmetrise(prob):
    """get_edge_prediction, MolNexTR/components.py:383-400, on a [k,k,7] float64 array (the reference works on
    Python float lists = float64). Classes 0-4 are averaged with the transpose; 5 (solid wedge) and 6 (dashed
    wedge) are cross-averaged: e[i][j][5] <- (e[i][j][5]+e[j][i][6])/2, e[i][j][6] <- (e[i][j][6]+e[j][i][5])/2,
    then e[j][i][5] <- e[i][j][6], e[j][i][6] <- e[i][j][5]. Only i<j pairs are rewritten; the diagonal is untouched."""
    e = np.array(prob, dtype=np.float64, copy=True)
    k = e.shape[0]
    if k == 0:
        return e
    iu, ju = np.triu_indices(k, 1)
    sym = (e[iu, ju, :5] + e[ju, iu, :5]) / 2
    w5 = (e[iu, ju, 5] + e[ju, iu, 6]) / 2
    w6 = (e[iu, ju, 6] + e[ju, iu, 5]) / 2
    e[iu, ju, :5] = sym
    e[ju, iu, :5] = sym
    e[iu, ju, 5], e[iu, ju, 6] = w5, w6
    e[ju, iu, 5], e[ju, iu, 6] = w6, w5
    return e


@torch.no_grad()
def predict_edges(hidden, atom_idx, sd):
    """The per-sample body of Decoder.decode's 'edges' branch, MolNexTR/components.py:478-484.
    Returns (edge class [k,k] int, edge score [k,k] float64)."""
    if len(atom_idx) == 0:
        return np.zeros((0, 0), dtype=np.int64), np.zeros((0, 0))
    prob = F.softmax(edge_logits(hidden, atom_idx, sd), dim=2).tolist()   # float32 -> python floats
    e = symmetrise(prob)
    return np.argmax(e, axis=2), np.max(e, axis=2)
