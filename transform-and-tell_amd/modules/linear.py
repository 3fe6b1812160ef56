"""tell/modules/linear.py:8-50 on the MI355X path."""
import math

import torch
import torch.nn as nn

from .. import ops


class GehringLinear(nn.Module):
    """Weight-normalised Linear with Gehring initialisation (tell/modules/linear.py:8-33).
    Parameters `weight_g` [out,1], `weight_v` [out,in], `bias` [out] exactly as
    torch.nn.utils.weight_norm(dim=0) registers them."""

    def __init__(self, in_features, out_features, dropout=0, bias=True, weight_norm=True):
        super().__init__()
        assert weight_norm, 'the hot path only instantiates weight-normalised GehringLinear'
        self.in_features, self.out_features, self.dropout = in_features, out_features, dropout
        std = math.sqrt((1 - dropout) / in_features)
        v = torch.randn(out_features, in_features) * std
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(v)

    def forward(self, x, act=0):
        """act: 0 none, 1 relu fused into the GEMM epilogue (FFN fc1)."""
        return ops.wn_linear(x, self.weight_g, self.weight_v, self.bias, act)

    def extra_repr(self):
        return 'in_features=%d, out_features=%d' % (self.in_features, self.out_features)


class Linear(nn.Module):
    """Plain xavier-initialised linear (`.weight`, optional `.bias`)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)
