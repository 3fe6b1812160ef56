"""tell/modules/token_embedders/{adaptive,positional,sum_text_field_embedder}.py on the MI355X path."""
import math

import torch
import torch.nn as nn

from .. import ops
from .. import runtime as rt
from ..common.registrable import Registrable


class TokenEmbedder(nn.Module, Registrable):
    def get_output_dim(self):
        raise NotImplementedError


class TextFieldEmbedder(nn.Module, Registrable):
    pass


def make_positions(X, padding_idx, left_pad=False, onnx_trace=False):
    """positional.py:231-268 (integer index arithmetic on the host side of the boundary)."""
    n = X.shape[1]
    pos = torch.arange(padding_idx + 1, padding_idx + 1 + n, dtype=X.dtype, device=X.device)[None, :].expand_as(X)
    mask = X.ne(padding_idx)
    if left_pad:
        pos = pos - (n - mask.long().sum(dim=1, keepdim=True))
    return torch.where(mask, pos, torch.full_like(X, padding_idx))


def sinusoid_table(n_rows, dim, padding_idx):
    """positional.py:126-165 - a constant buffer built once on the host."""
    half = dim // 2
    inc = math.log(10000.0) / (half - 1)
    inv = torch.exp(torch.arange(half, dtype=torch.float) * -inc)
    ang = torch.arange(n_rows, dtype=torch.float)[:, None] * inv[None, :]
    tab = torch.cat([ang.sin(), ang.cos()], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n_rows, 1)], dim=1)
    if padding_idx is not None:
        tab[padding_idx] = 0
    return tab


@TokenEmbedder.register('adaptive')
class AdaptiveEmbedding(TokenEmbedder):
    """adaptive.py:12-80.  `embeddings.i.0.weight` band tables, `embeddings.i.1.weight` projections."""

    def __init__(self, vocab=None, namespace=None, padding_idx=0, initial_dim=1024, factor=1., output_dim=1024,
                 cutoff=(), vocab_size=None, scale_embeds=False):
        super().__init__()
        vocab_size = vocab_size or vocab.get_vocab_size(namespace)
        cutoff = list(cutoff)
        if not cutoff or vocab_size > cutoff[-1]:
            cutoff.append(vocab_size)
        assert vocab_size == cutoff[-1]
        self.cutoff, self.embed_size, self.padding_idx = cutoff, output_dim, padding_idx
        self.embed_scale = math.sqrt(output_dim) if scale_embeds else 1
        self.embeddings = nn.ModuleList()
        prev = 0
        for i, c in enumerate(cutoff):
            dim = int(initial_dim // (factor ** i))
            emb = nn.Embedding(c - prev, dim, padding_idx)
            emb.weight.data.normal_(0, math.sqrt(1 / dim))
            emb.weight.data[padding_idx].zero_()
            proj = nn.Linear(dim, output_dim, bias=False)
            nn.init.xavier_uniform_(proj.weight)
            self.embeddings.append(nn.Sequential(emb, proj))
            prev = c

    def weights_for_band(self, band):
        return self.embeddings[band][0].weight, self.embeddings[band][1].weight

    def get_output_dim(self):
        return self.embed_size

    def tables(self):
        out = []
        for s in self.embeddings:
            out += [s[0].weight, s[1].weight]
        return out


@TokenEmbedder.register('sinusoidal_positional')
class SinusoidalPositionalEmbedding(TokenEmbedder):
    """positional.py:85-229 (buffer `weights`); the lookup itself is fused into the
    embedding kernel by SumTextFieldEmbedder."""

    _instances = [0]

    def __init__(self, vocab=None, embedding_dim=1024, padding_idx=1, left_pad=False, init_size=1024):
        super().__init__()
        if left_pad:
            raise NotImplementedError('caption decoding uses right padding (left_pad: false in every config)')
        self.embedding_dim, self.padding_idx, self.left_pad = embedding_dim, padding_idx, left_pad
        self.register_buffer('weights', sinusoid_table(init_size + 1, embedding_dim, padding_idx))
        SinusoidalPositionalEmbedding._instances[0] += 1
        self._state_key = 'SinusoidalPositionalEmbedding.%d.position' % SinusoidalPositionalEmbedding._instances[0]

    def get_output_dim(self):
        return self.embedding_dim

    def next_start(self, n, incremental_state):
        """positional.py:170-187: position offset of this call + table growth."""
        start = 0
        if incremental_state is not None:
            start = incremental_state.get(self._state_key, 0)
            incremental_state[self._state_key] = start + n
        need = start + n + self.padding_idx + 1
        if need > self.weights.shape[0]:
            self.weights = sinusoid_table(need, self.embedding_dim, self.padding_idx).to(self.weights.device)
        return start


@TextFieldEmbedder.register('sum')
class SumTextFieldEmbedder(TextFieldEmbedder):
    """sum_text_field_embedder.py:16-163 specialised to the hot path's pair
    {adaptive, position}: one fused HIP pipeline produces scale*adaptive + sinusoid."""

    def __init__(self, token_embedders, embedder_to_indexer_map=None, allow_unmatched_keys=False):
        super().__init__()
        self._keys = sorted(token_embedders)
        self._map = embedder_to_indexer_map
        for k, m in token_embedders.items():
            self.add_module('token_embedder_%s' % k, m)
        if set(self._keys) != {'adaptive', 'position'}:
            raise NotImplementedError('the HIP embedder implements the {adaptive, position} sum of the configs')

    def get_output_dim(self):
        return max(getattr(self, 'token_embedder_%s' % k).get_output_dim() for k in self._keys)

    def forward(self, text_field_input, num_wrapping_dims=0, incremental_state=None):
        ad, po = self.token_embedder_adaptive, self.token_embedder_position
        src = self._map['adaptive'][0] if self._map is not None else 'adaptive'
        ids = text_field_input[src]
        B, T = ids.shape
        start = po.next_start(T, incremental_state)
        from .. import decode
        if decode.embed_usable(self, ids, incremental_state):          # generation step: two launches (decode.py)
            return decode.embed_step(self, ids.contiguous(), start).transpose(0, 1)
        tbc = ops.adaptive_embed(ids, po.weights, ad.cutoff, ad.embed_scale, po.padding_idx, start,
                                 ad.padding_idx, ad.tables())
        return tbc.transpose(0, 1)            # [B,T,E] view of the decoder's T x B x C buffer
