import torch.nn as nn
from .layers import trunc_normal_


def checkpoint_filter_fn(state_dict, model):
    return state_dict


def _init_vit_weights(m, n='', head_bias=0., jax_impl=False):
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)
