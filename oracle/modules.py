"""nn.Module containers for the oracle, with the reference's class names,
constructor arguments and `state_dict` keys so that
weights move freely between reference <-> oracle <-> HIP product.

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as OF


def _maybe_dropout(x, p, training):
    return F.dropout(x, p=p, training=training) if (training and p > 0) else x


class GehringLinear(nn.Module):
    """tell/modules/linear.py:8-33 - params weight_g [out,1], weight_v [out,in], bias."""

    def __init__(self, in_features, out_features, dropout=0.0, bias=True):
        super().__init__()
        std = math.sqrt((1 - dropout) / in_features)
        v = torch.randn(out_features, in_features) * std
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(v)

    def forward(self, x):
        return OF.gehring_linear(x, self.weight_g, self.weight_v, self.bias)


class _PlainLinear(nn.Module):
    """xavier-uniform Linear holder giving a `.weight` (and optional `.bias`) key."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class DynamicConv1dTBC(nn.Module):
    """tell/modules/convolutions/dynamic.py:25-361 as configured by the decoders
    (weight_softmax=True, padding_l=K-1, no bias, no in_proj, no renorm)."""

    _instances = 0

    def __init__(self, input_size, kernel_size, num_heads, weight_dropout=0.0):
        super().__init__()
        self.input_size, self.kernel_size, self.num_heads = input_size, kernel_size, num_heads
        self.weight_dropout = weight_dropout
        self.weight_linear = _PlainLinear(input_size, num_heads * kernel_size, bias=False)
        DynamicConv1dTBC._instances += 1
        # same key convention as tell/utils/state.py:21-34
        self._state_key = 'DynamicConv1dTBC.%d.input_buffer' % DynamicConv1dTBC._instances

    def forward(self, x, incremental_state=None, drop_mask=None):
        n_hist = 0
        if incremental_state is not None:                               # dynamic.py:94-99
            prev = incremental_state.get(self._state_key)
            if prev is not None:
                n_hist = prev.shape[0]
                x = torch.cat([prev, x], dim=0)
            incremental_state[self._state_key] = x[-self.kernel_size + 1:] if self.kernel_size > 1 \
                else x[:0]
        logits = self.weight_linear(x)
        p = self.weight_dropout if self.training else 0.0
        if drop_mask is None and p > 0:
            drop_mask = (torch.rand(x.shape[0], x.shape[1], self.num_heads, self.kernel_size) >= p).float()
        taps = OF.dynamic_conv_taps(logits, self.num_heads, self.kernel_size,
                                    drop_mask if p > 0 else None, p)
        out = OF.dynamic_conv_apply(x, taps)
        return out[n_hist:]                                              # dynamic.py:115-116


class LightweightConv1dTBC(nn.Module):
    """tell/modules/convolutions/lightweight.py:83-240 as `decoder_conv_type: lightweight` builds it
    (decoder_faces_objects.py:199-203: weight_softmax, padding_l = K-1, no bias): one learned tap vector per head,
    softmax over ALL K taps (:163-164 / :186-187), DropConnect on the taps (:174 / :204-205), then the same causal
    K-tap sum as the dynamic convolution - a window shorter than K simply drops the taps that reach before the
    start (:166-168 incremental, :193-195 K > T)."""

    _instances = 0

    def __init__(self, input_size, kernel_size, num_heads, weight_dropout=0.0):
        super().__init__()
        self.input_size, self.kernel_size, self.num_heads = input_size, kernel_size, num_heads
        self.weight_dropout = weight_dropout
        self.weight = nn.Parameter(torch.empty(num_heads, 1, kernel_size))
        nn.init.xavier_uniform_(self.weight)                                # lightweight.py:127
        LightweightConv1dTBC._instances += 1
        self._state_key = 'LightweightConv1dTBC.%d.input_buffer' % LightweightConv1dTBC._instances

    def forward(self, x, incremental_state=None, drop_mask=None):
        n_hist = 0
        if incremental_state is not None:                               # lightweight.py:152-160
            prev = incremental_state.get(self._state_key)
            if prev is not None:
                n_hist = prev.shape[0]
                x = torch.cat([prev, x], dim=0)
            incremental_state[self._state_key] = x[-self.kernel_size + 1:] if self.kernel_size > 1 \
                else x[:0]
        T, B, _ = x.shape
        logits = self.weight.view(1, 1, -1).expand(T, B, self.num_heads * self.kernel_size)
        p = self.weight_dropout if self.training else 0.0
        if drop_mask is None and p > 0:
            drop_mask = (torch.rand(T, B, self.num_heads, self.kernel_size) >= p).float()
        taps = OF.dynamic_conv_taps(logits, self.num_heads, self.kernel_size, drop_mask if p > 0 else None, p)
        return OF.dynamic_conv_apply(x, taps)[n_hist:]


class MultiHeadAttention(nn.Module):
    """tell/modules/attention/multi_head.py:207-552 (add_bias_kv, add_zero_attn)."""

    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.kdim = embed_dim if kdim is None else kdim
        self.vdim = embed_dim if vdim is None else vdim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        E = embed_dim
        if self.qkv_same_dim:
            self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
            nn.init.xavier_uniform_(self.in_proj_weight)
        else:
            self.k_proj_weight = nn.Parameter(torch.empty(E, self.kdim))
            self.v_proj_weight = nn.Parameter(torch.empty(E, self.vdim))
            self.q_proj_weight = nn.Parameter(torch.empty(E, E))
            for w in (self.k_proj_weight, self.v_proj_weight, self.q_proj_weight):
                nn.init.xavier_uniform_(w)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        self.out_proj = _PlainLinear(E, E, bias=True)
        self.bias_k = nn.Parameter(torch.empty(1, 1, E))
        self.bias_v = nn.Parameter(torch.empty(1, 1, E))
        nn.init.xavier_normal_(self.bias_k)
        nn.init.xavier_normal_(self.bias_v)

    def qkv_weights(self):
        E = self.embed_dim
        if self.qkv_same_dim:
            w = self.in_proj_weight
            return w[:E], w[E:2 * E], w[2 * E:]
        return self.q_proj_weight, self.k_proj_weight, self.v_proj_weight

    def forward(self, query, key, key_padding_mask=None, need_weights=False, drop_mask=None):
        wq, wk, wv = self.qkv_weights()
        p = self.dropout if self.training else 0.0
        if drop_mask is None and p > 0:
            S1 = (key.shape[0] if key.shape[2] > 0 else 0) + 2
            drop_mask = (torch.rand(query.shape[1] * self.num_heads, query.shape[0], S1) >= p).float()
        return OF.cross_attention(query, key, key_padding_mask, wq, wk, wv, self.in_proj_bias,
                                  self.bias_k, self.bias_v, self.out_proj.weight,
                                  self.out_proj.bias, self.num_heads, need_weights,
                                  drop_mask if p > 0 else None, p)


class AdaptiveEmbedding(nn.Module):
    """tell/modules/token_embedders/adaptive.py:12-80.  `embeddings.i.0.weight`
    (band table, row `padding_idx` zero and gradient-free, :42,:51) and
    `embeddings.i.1.weight` (bias-free projection)."""

    def __init__(self, vocab_size, padding_idx, initial_dim, factor, output_dim, cutoff,
                 scale_embeds=False):
        super().__init__()
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

    def forward(self, ids, incremental_state=None):
        # nn.Embedding(padding_idx=0) semantics: the lookup path never sends
        # gradient into local row 0 of a band -> detach those rows on lookup.
        embs = []
        for seq in self.embeddings:
            w = seq[0].weight
            keep = torch.ones(w.shape[0], 1)
            keep[self.padding_idx] = 0
            embs.append(w * keep + (w * (1 - keep)).detach())
        return OF.adaptive_embed(ids, self.cutoff, embs,
                                 [s[1].weight for s in self.embeddings], self.embed_scale)


class SinusoidalPositionalEmbedding(nn.Module):
    """tell/modules/token_embedders/positional.py:85-229 (buffer `weights`)."""

    _instances = 0

    def __init__(self, embedding_dim, padding_idx, left_pad, init_size=1024):
        super().__init__()
        self.embedding_dim, self.padding_idx, self.left_pad = embedding_dim, padding_idx, left_pad
        self.register_buffer('weights', OF.sinusoid_table(init_size + 1, embedding_dim, padding_idx))
        SinusoidalPositionalEmbedding._instances += 1
        self._state_key = 'SinusoidalPositionalEmbedding.%d.position' % SinusoidalPositionalEmbedding._instances

    def get_output_dim(self):
        return self.embedding_dim

    def forward(self, ids, incremental_state=None):
        n = ids.shape[1]
        start = 0
        if incremental_state is not None:                                # positional.py:170-173
            start = incremental_state.get(self._state_key, 0)
            incremental_state[self._state_key] = start + n
        need = start + n + 1
        if need > self.weights.shape[0]:                                 # :180-187
            self.weights = OF.sinusoid_table(need, self.embedding_dim, self.padding_idx)
        pos = OF.make_positions(ids, self.padding_idx, self.left_pad)
        pos = torch.where(pos != self.padding_idx, pos + start, pos)
        return self.weights[pos].detach()


class SumTextFieldEmbedder(nn.Module):
    """tell/modules/token_embedders/sum_text_field_embedder.py:16-118: sub-modules
    `token_embedder_<key>`; outputs summed (keys visited in sorted order)."""

    def __init__(self, token_embedders, embedder_to_indexer_map=None, allow_unmatched_keys=False):
        super().__init__()
        self._keys = sorted(token_embedders)
        self._map = embedder_to_indexer_map
        for k, m in token_embedders.items():
            self.add_module('token_embedder_%s' % k, m)

    def get_output_dim(self):
        return max(getattr(self, 'token_embedder_%s' % k).get_output_dim() for k in self._keys)

    def forward(self, text_field_input, incremental_state=None):
        total = None
        for k in self._keys:
            src = self._map[k][0] if self._map is not None else k
            v = getattr(self, 'token_embedder_%s' % k)(text_field_input[src],
                                                       incremental_state=incremental_state)
            total = v if total is None else total + v
        return total


class _TiedLinear(nn.Module):
    """tell/modules/linear.py:37-50 - registers the shared Parameter as `.weight`."""

    def __init__(self, weight):
        super().__init__()
        self.weight = weight


class _TiedHead(nn.Module):
    """tell/modules/softmax.py:11-40."""

    def __init__(self, tied_emb, input_dim, n_classes):
        super().__init__()
        self.word_proj = _TiedLinear(tied_emb)
        self.class_proj = _PlainLinear(input_dim, n_classes, bias=False)
        self.register_buffer('_float_tensor', torch.FloatTensor(1).zero_())


class AdaptiveSoftmax(nn.Module):
    """tell/modules/softmax.py:43-222 as configured (tie_adaptive_weights=True,
    tie_adaptive_proj=False, factor 1, dropout 0)."""

    def __init__(self, vocab_size, input_dim, cutoff, adaptive_inputs):
        super().__init__()
        cutoff = list(cutoff)
        if not cutoff or vocab_size > cutoff[-1]:
            cutoff.append(vocab_size)
        self.cutoff, self.vocab_size, self.input_dim = cutoff, vocab_size, input_dim
        n_tails = len(cutoff) - 1
        self.head = _TiedHead(adaptive_inputs.weights_for_band(0)[0], input_dim, n_tails)
        self.tail = nn.ModuleList()
        for i in range(n_tails):
            emb, proj = adaptive_inputs.weights_for_band(i + 1)
            self.tail.append(nn.Sequential(_PlainLinear(input_dim, proj.shape[1], bias=False),
                                           nn.Dropout(0.0), _TiedLinear(emb)))
        self.register_buffer('version', torch.LongTensor([1]))

    def _weights(self):
        return (self.head.word_proj.weight, self.head.class_proj.weight,
                [t[0].weight for t in self.tail], [t[2].weight for t in self.tail])

    def get_log_prob(self, x, target=None):
        B, T, E = x.shape
        return OF.adaptive_log_probs(x.reshape(-1, E), self.cutoff, *self._weights()).view(B, T, -1)

    def loss(self, x, target, padding_idx):
        return OF.adaptive_loss_sum(x, target, self.cutoff, *self._weights(), padding_idx=padding_idx)


class AdaptiveLoss(nn.Module):
    """tell/modules/criteria/adaptive_loss.py:11-73."""

    def __init__(self, padding_idx=1):
        super().__init__()
        self.padding_idx = padding_idx

    def forward(self, adaptive_softmax, net_output, decoder_target):
        return adaptive_softmax.loss(net_output[0], decoder_target, self.padding_idx)
