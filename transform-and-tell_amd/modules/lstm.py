"""LSTM-decoder building blocks of the GloVe/LSTM baseline (tell/models/decoder_flattened_lstm.py:20-65) on the MI355X
path: the gate GEMMs are `ops.linear` (MFMA GEMM, fused bias), the cell update / attention core / tanh are the kernels
of csrc/lstm.hip.  Parameter names are nn.LSTMCell's and the reference AttentionLayer's, so checkpoints load."""
import torch
import torch.nn as nn

from .. import ops
from .linear import GehringLinear


class LSTMCell(nn.Module):
    """`LSTMCell()` of decoder_flattened_lstm.py:20-26: nn.LSTMCell with uniform(-0.1, 0.1) parameters."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-0.1, 0.1))
        self.weight_hh = nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-0.1, 0.1))
        self.bias_ih = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-0.1, 0.1))
        self.bias_hh = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-0.1, 0.1))

    def forward(self, x, state):
        h, c = state
        g1 = ops.linear(x, self.weight_ih, self.bias_ih)
        g2 = ops.linear(h, self.weight_hh, self.bias_hh)
        return ops.lstm_cell(g1, g2, c)


class AttentionLayer(nn.Module):
    """decoder_flattened_lstm.py:29-65."""

    def __init__(self, input_embed_dim, source_embed_dim, output_embed_dim, bias=False):
        super().__init__()
        self.input_proj = GehringLinear(input_embed_dim, source_embed_dim, bias=bias)
        self.output_proj = GehringLinear(input_embed_dim + source_embed_dim, output_embed_dim, bias=bias)

    def forward(self, input, source_hids, encoder_padding_mask):
        x = self.input_proj(input)                                       # bsz x source_embed_dim
        mask = encoder_padding_mask
        if mask is not None and mask.dtype != torch.uint8:
            mask = mask.to(torch.uint8)
        ctx, probs = ops.dot_attention(x, source_hids, mask.contiguous() if mask is not None else None)   # :46-60
        return ops.tanh(self.output_proj(torch.cat((ctx, input), dim=1))), probs      # :62
