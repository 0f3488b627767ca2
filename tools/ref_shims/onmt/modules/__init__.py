"""OpenNMT-py 2.2.0 MultiHeadedAttention restated from its documented algorithm.

q,k,v = Linear(256,256) each; heads are contiguous 32-wide channel slices;
q is divided by sqrt(d_head) BEFORE QK^T; scores are promoted to fp32;
masked positions are filled with -1e18; softmax; dropout; P.V; final_linear.
Stepwise decoding: self-attention K/V are concatenated along the key axis of
[B,h,t,d]; cross-attention K/V are projected once and cached.
"""
import math
import torch
import torch.nn as nn


class AverageAttention(nn.Module):
    pass


class MultiHeadedAttention(nn.Module):
    def __init__(self, head_count, model_dim, dropout=0.1, max_relative_positions=0):
        assert model_dim % head_count == 0 and max_relative_positions == 0
        super().__init__()
        self.dim_per_head = model_dim // head_count
        self.model_dim = model_dim
        self.head_count = head_count
        self.linear_keys = nn.Linear(model_dim, model_dim)
        self.linear_values = nn.Linear(model_dim, model_dim)
        self.linear_query = nn.Linear(model_dim, model_dim)
        self.softmax = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.final_linear = nn.Linear(model_dim, model_dim)

    def _split(self, x, b):
        return x.view(b, -1, self.head_count, self.dim_per_head).transpose(1, 2)

    def forward(self, key, value, query, mask=None, layer_cache=None, attn_type=None):
        b = key.size(0)
        if layer_cache is not None and attn_type == "self":
            q, k, v = self.linear_query(query), self.linear_keys(query), self.linear_values(query)
            k, v = self._split(k, b), self._split(v, b)
            if layer_cache["self_keys"] is not None:
                k = torch.cat((layer_cache["self_keys"], k), dim=2)
            if layer_cache["self_values"] is not None:
                v = torch.cat((layer_cache["self_values"], v), dim=2)
            layer_cache["self_keys"], layer_cache["self_values"] = k, v
        elif layer_cache is not None and attn_type == "context":
            q = self.linear_query(query)
            if layer_cache["memory_keys"] is None:
                k, v = self._split(self.linear_keys(key), b), self._split(self.linear_values(value), b)
            else:
                k, v = layer_cache["memory_keys"], layer_cache["memory_values"]
            layer_cache["memory_keys"], layer_cache["memory_values"] = k, v
        else:
            k, v = self._split(self.linear_keys(key), b), self._split(self.linear_values(value), b)
            q = self.linear_query(query)
        q = self._split(q, b) / math.sqrt(self.dim_per_head)
        scores = torch.matmul(q, k.transpose(2, 3)).float()
        if mask is not None:
            scores = scores.masked_fill(mask.unsqueeze(1), -1e18)
        attn = self.softmax(scores).to(q.dtype)
        ctx = torch.matmul(self.dropout(attn), v)
        ctx = ctx.transpose(1, 2).contiguous().view(b, -1, self.model_dim)
        return self.final_linear(ctx), attn

    def update_dropout(self, dropout):
        self.dropout.p = dropout
