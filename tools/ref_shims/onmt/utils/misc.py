import torch


def sequence_mask(lengths, max_len=None):
    max_len = max_len or int(lengths.max())
    return torch.arange(0, max_len, device=lengths.device)[None, :] < lengths[:, None]
