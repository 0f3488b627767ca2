"""OpenNMT-py 2.2.0 Elementwise restated: split the feature dim, apply one module per feature, merge."""
import torch
import torch.nn as nn


class Elementwise(nn.ModuleList):
    def __init__(self, merge=None, *args):
        assert merge in (None, 'first', 'concat', 'sum', 'mlp')
        self.merge = merge
        super().__init__(*args)

    def forward(self, emb):
        feats = [f.squeeze(2) for f in emb.split(1, dim=2)]
        assert len(self) == len(feats)
        outs = [m(x) for m, x in zip(self, feats)]
        if self.merge == 'first':
            return outs[0]
        if self.merge in ('concat', 'mlp'):
            return torch.cat(outs, 2)
        if self.merge == 'sum':
            return sum(outs)
        return outs
