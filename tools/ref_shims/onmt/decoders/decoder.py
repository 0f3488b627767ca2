import torch.nn as nn


class DecoderBase(nn.Module):
    def __init__(self, attentional=True):
        super().__init__()
        self.attentional = attentional
