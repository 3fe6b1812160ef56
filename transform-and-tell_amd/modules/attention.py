"""tell/modules/attention/multi_head.py:207-552 on the MI355X path."""
import torch
import torch.nn as nn

from .. import ops
from .linear import Linear


class MultiHeadAttention(nn.Module):
    """Cross attention with separate key/value dims, learned bias_k/bias_v row and the
    extra zero-attention row; called by the decoders with static_kv=True and
    incremental_state=None (decoder_faces_objects.py:275-282)."""

    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0., bias=True, add_bias_kv=True,
                 add_zero_attn=True, self_attention=False, encoder_decoder_attention=False, out_dim=None):
        super().__init__()
        if not bias or self_attention or encoder_decoder_attention or out_dim is not None:
            raise NotImplementedError('only the configuration used by the caption decoders is implemented')
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.kdim = embed_dim if kdim is None else kdim
        self.vdim = embed_dim if vdim is None else vdim
        assert self.kdim == self.vdim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
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
        self.out_proj = Linear(E, E, bias=True)
        self.add_zero_attn = add_zero_attn
        if add_bias_kv:
            self.bias_k = nn.Parameter(torch.empty(1, 1, E))
            self.bias_v = nn.Parameter(torch.empty(1, 1, E))
            nn.init.xavier_normal_(self.bias_k)
            nn.init.xavier_normal_(self.bias_v)
        else:
            self.bias_k = self.bias_v = None

    def _wrows(self, i):
        E = self.embed_dim
        if self.qkv_same_dim:
            return self.in_proj_weight, (i * E, (i + 1) * E)
        return (self.q_proj_weight, self.k_proj_weight, self.v_proj_weight)[i], None

    def project_kv(self, key, key_t=None):
        """K and V projections of a context [S,B,kdim] -> two [S,B,E] views.  When the context is
        stored batch-major (RoBERTa / ResNet outputs) the GEMM runs on that storage directly."""
        E = self.embed_dim
        if key.shape[2] == 0 or key.shape[0] == 0:              # empty context, multi_head.py:349-374
            z = key.new_zeros(0, key.shape[1], E)
            return z, z
        bm = key.transpose(0, 1)
        src = bm if (bm.is_contiguous() and not key.is_contiguous()) else key.contiguous()
        wk, rk = self._wrows(1)
        wv, rv = self._wrows(2)
        k = ops.linear(src, wk, self.in_proj_bias, rows=rk, b_rows=(E, 2 * E), x_t=key_t)
        v = ops.linear(src, wv, self.in_proj_bias, rows=rv, b_rows=(2 * E, 3 * E), x_t=key_t)
        if src is bm:
            k, v = k.transpose(0, 1), v.transpose(0, 1)
        return k, v

    def project_kv_packed(self, key):
        """K and V of a (non-empty) context as ONE projection [S,B,2E] (ops.kv_linear): half the launches, twice the
        N of the GEMMs forward and backward; the attention kernels read the halves through their strides."""
        E = self.embed_dim
        bm = key.transpose(0, 1)
        src = bm if (bm.is_contiguous() and not key.is_contiguous()) else key.contiguous()
        wk, rk = self._wrows(1)
        wv, rv = self._wrows(2)
        kv = ops.kv_linear(src, wk, rk, wv, rv, self.in_proj_bias, E)
        return kv.transpose(0, 1) if src is bm else kv

    # The three stages of forward() for a caller that runs SEVERAL attentions side by side (the decoder's context
    # block): the query and output projections of all of them go out as one grouped launch each (ops.grouped_linear).
    def q_spec(self):
        wq, rq = self._wrows(0)
        return (wq, rq, self.in_proj_bias, (0, self.embed_dim), self.scaling)                      # :348-353

    def out_spec(self):
        return (self.out_proj.weight, None, self.out_proj.bias, None, 1.0)

    def core(self, q, key, key_padding_mask=None, kv=None, packed=None):
        """softmax(q K^T) V between the two projections: q already projected and scaled; K / V projected here (one
        packed projection when gradients are on), taken from `kv` (generation: projected once per caption) or from
        `packed` (training: the [S,B,2E] projection blocks.kv_project_all computed for all layers at once)."""
        T, B, E = q.shape
        if packed is not None:
            k = v = packed
        elif kv is None and ops.rt.compute_dtype() == torch.bfloat16 and key.shape[0] > 0 and key.shape[2] > 0 \
                and torch.is_grad_enabled():
            packed = self.project_kv_packed(key)          # training: K and V as one [S,B,2E] projection
            k = v = packed
        else:
            k, v = kv if kv is not None else self.project_kv(key)
        mask = None
        if key_padding_mask is not None and k.shape[0] > 0:
            mask = key_padding_mask if key_padding_mask.dtype == torch.uint8 else \
                key_padding_mask.to(torch.uint8).contiguous()
        beams = 1
        if kv is not None and T == 1 and k.shape[0] > 0 and B != k.shape[1] and B % k.shape[1] == 0:
            # beam search: the n hypotheses of a sample attend to the SAME static context (see forward())
            beams = B // k.shape[1]
            q = q.view(k.shape[1], beams, E).transpose(0, 1)                      # [n, B/n, E] strided view
        attn = ops.attention(q, k, None if packed is not None else v, mask, self.bias_k, self.bias_v, self.num_heads,
                             self.add_zero_attn, self.dropout, self.training)
        if beams > 1:
            attn = attn.transpose(0, 1).reshape(1, B, E)
        return attn

    def forward(self, query, key, value=None, key_padding_mask=None, incremental_state=None,
                need_weights=True, static_kv=True, attn_mask=None, key_t=None, kv=None):
        """kv: optional (k, v) already projected by `project_kv` - the contexts are static during
        generation, so they are projected once per caption instead of once per generated token."""
        assert attn_mask is None and incremental_state is None
        T, B, E = query.shape
        assert E == self.embed_dim
        wq, rq = self._wrows(0)
        q = ops.linear(query, wq, self.in_proj_bias, rows=rq, b_rows=(0, E), alpha=self.scaling)  # :348-353
        packed = None
        if kv is None and ops.rt.compute_dtype() == torch.bfloat16 and key.shape[0] > 0 and key.shape[2] > 0 \
                and torch.is_grad_enabled():
            packed = self.project_kv_packed(key)          # training: K and V as one [S,B,2E] projection
            k = v = packed
        else:
            k, v = kv if kv is not None else self.project_kv(key, key_t)
        mask = None
        if key_padding_mask is not None and k.shape[0] > 0:
            # (the decoder converts every context mask once per step - 4 layers share it)
            mask = key_padding_mask if key_padding_mask.dtype == torch.uint8 else \
                key_padding_mask.to(torch.uint8).contiguous()
        beams = 1
        if kv is not None and T == 1 and k.shape[0] > 0 and B != k.shape[1] and B % k.shape[1] == 0:
            # beam search: the n hypotheses of a sample (rows b*n + j) attend to the SAME static context, so they are
            # presented to the kernel as n query positions of batch element b - K/V are read once per sample, not
            # once per hypothesis (cross-attention has no causal mask, so query positions are independent)
            beams = B // k.shape[1]
            q = q.view(k.shape[1], beams, E).transpose(0, 1)                      # [n, B/n, E] strided view
        attn, lse = ops.attention(q, k, None if packed is not None else v, mask, self.bias_k, self.bias_v,
                                  self.num_heads, self.add_zero_attn, self.dropout, self.training, return_lse=True)
        if packed is not None:
            k = packed[..., :E]
        if beams > 1:
            assert not need_weights
            attn = attn.transpose(0, 1).reshape(1, B, E)
        out = self.out_proj(attn)
        weights = None
        if need_weights:
            weights = ops.attention_avg_weights(q.detach(), k.detach(), mask, self.bias_k, lse, self.num_heads,
                                                self.add_zero_attn)
        return out, weights
