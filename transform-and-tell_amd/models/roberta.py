"""RoBERTa-large article encoder (fairseq `roberta.large`, loaded by the reference with
torch.hub at transformer_faces_objects.py:49-50 and called at :352-353) on the MI355X path.

24 post-LN blocks of {fused QKV GEMM, MFMA self-attention with key-padding mask, out-proj
GEMM, residual+LayerNorm, fc1 GEMM with fused bias+erf-GELU, fc2 GEMM, residual+LayerNorm};
rows are kept batch-major [B*S, E] and every layer's output is written straight into the
[25, B, S, E] stack that `extract_features(..., return_all_hiddens=True)` returns.  Dropout
runs when `self.training` (the reference leaves the frozen encoder in train mode)."""
import os

import torch
import torch.nn as nn

from .. import hip, ops
from .. import runtime as rt

call = hip.call


class _SelfAttn(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E).normal_(0, 0.02))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        self.out_proj = nn.Linear(E, E)
        nn.init.normal_(self.out_proj.weight, 0, 0.02)
        nn.init.zeros_(self.out_proj.bias)


class _EncLayer(nn.Module):
    def __init__(self, E, FF):
        super().__init__()
        self.self_attn = _SelfAttn(E)
        self.self_attn_layer_norm = nn.LayerNorm(E)
        self.fc1 = nn.Linear(E, FF)
        self.fc2 = nn.Linear(FF, E)
        self.final_layer_norm = nn.LayerNorm(E)
        for l in (self.fc1, self.fc2):
            nn.init.normal_(l.weight, 0, 0.02)
            nn.init.zeros_(l.bias)


class _SentenceEncoder(nn.Module):
    def __init__(self, V, E, FF, L, max_pos, pad):
        super().__init__()
        self.embed_tokens = nn.Embedding(V, E, pad)
        self.embed_positions = nn.Embedding(max_pos + pad + 1, E, pad)
        nn.init.normal_(self.embed_tokens.weight, 0, 0.02)
        nn.init.normal_(self.embed_positions.weight, 0, 0.02)
        self.embed_tokens.weight.data[pad].zero_()
        self.embed_positions.weight.data[pad].zero_()
        self.emb_layer_norm = nn.LayerNorm(E)
        self.layers = nn.ModuleList([_EncLayer(E, FF) for _ in range(L)])


class RobertaEncoder(nn.Module):
    def __init__(self, vocab=50265, dim=1024, ffn=4096, layers=24, heads=16, max_positions=512, pad=1,
                 dropout=0.1, attention_dropout=0.1):
        super().__init__()
        self.dim, self.heads, self.pad, self.n_layers = dim, heads, pad, layers
        self.dropout, self.attention_dropout = dropout, attention_dropout
        self.model = nn.Module()
        self.model.decoder = nn.Module()
        self.model.decoder.sentence_encoder = _SentenceEncoder(vocab, dim, ffn, layers, max_positions, pad)

    def _qkv(self, attn):
        """fused [3E, E] weight / [3E] bias with the q rows pre-scaled by head_dim^-0.5 (= 1/8 for
        head_dim 64: exact in binary floating point, so identical to scaling q afterwards)."""
        E = self.dim
        s = (E // self.heads) ** -0.5

        def make():
            w = attn.in_proj_weight.detach().clone()
            b = attn.in_proj_bias.detach().clone()
            w[:E] *= s
            b[:E] *= s
            return ops.cast(w, rt.compute_dtype()), b
        return ops._cached(attn.in_proj_weight, ('qkv', attn.in_proj_bias._version), make)

    @staticmethod
    def _proj_residual_ln(inp, lin, residual, ln, scratch, out, M, E, p, fused):
        """out = LayerNorm(residual + dropout_p(lin(inp)))  (fairseq TransformerSentenceEncoderLayer, post-norm).  bf16 at
        shapes the resident 256x256 GEMM takes: the residual and the dropout ride in the GEMM epilogue
        (tell_gemm_nt_dropout_residual, same mask as the LayerNorm kernel would draw) and the LayerNorm reads one tensor;
        otherwise GEMM, then LayerNorm(dropout(x) + residual) as one launch."""
        w, b = ops.weight(lin.weight), lin.bias.detach()
        salt = rt.next_salt() if p > 0 else 0
        dcode = hip.dt(inp.dtype)
        if fused and hip.call_rc('tell_gemm_nt_dropout_residual', inp, inp.stride(0), w, w.stride(0), b, residual,
                                 residual.stride(0), scratch, scratch.stride(0), M, E, inp.shape[1], p, rt.seed(),
                                 salt) == 0:
            call('tell_layernorm_fwd', scratch, E, None, 0, ln.weight.detach(), ln.bias.detach(), out, E, None, None, M,
                 E, ln.eps, 0.0, 0, 0, dcode)
            return
        ops.gemm(inp, w, out=scratch, bias=b, bias_mode=1)
        call('tell_layernorm_fwd', scratch, E, residual, E, ln.weight.detach(), ln.bias.detach(), out, E, None, None, M, E,
             ln.eps, p, rt.seed(), salt, dcode)

    @torch.no_grad()
    def extract_features(self, ids, return_all_hiddens=False):
        hip.require_gpu()
        enc = self.model.decoder.sentence_encoder
        dtype = rt.compute_dtype()
        dcode = hip.dt(dtype)
        B, S = ids.shape
        E, H, M = self.dim, self.heads, B * S
        dev = ids.device
        tr = self.training
        p_h = self.dropout if tr else 0.0
        stack = torch.empty(self.n_layers + 1, B, S, E, dtype=dtype, device=dev)
        pad_mask = ids.eq(self.pad).to(torch.uint8).contiguous()
        pos_ws = torch.empty(B, S, dtype=torch.int32, device=dev)
        emb = torch.empty(M, E, dtype=dtype, device=dev)
        call('tell_roberta_embed', ids.contiguous(), B, S, self.pad, ops.weight(enc.embed_tokens.weight),
             ops.weight(enc.embed_positions.weight), pos_ws, emb, E, dcode)
        x = stack[0].view(M, E)
        ln = enc.emb_layer_norm
        if p_h > 0:
            lnout = torch.empty_like(emb)
            call('tell_layernorm_fwd', emb, E, None, 0, ln.weight.detach(), ln.bias.detach(), lnout, E, None, None, M,
                 E, ln.eps, 0.0, 0, 0, dcode)
            call('tell_dropout', lnout, x, M * E, p_h, rt.seed(), rt.next_salt(), dcode)
        else:
            call('tell_layernorm_fwd', emb, E, None, 0, ln.weight.detach(), ln.bias.detach(), x, E, None, None, M, E,
                 ln.eps, 0.0, 0, 0, dcode)
        call('tell_mask_rows', x, pad_mask, M, E, dcode)
        # (TELL_QKV_PAD=64: row stride 3E + 64 - alone the QKV GEMM gains 7 %, tools/probes/pp_ldc_sweep.py; in the step
        #  it measured neutral, 1469 / 1479 against 1478 / 1478 samples/s same box, so the packed layout stays)
        ldq = 3 * E + int(os.environ.get('TELL_QKV_PAD', '0'))
        qkv = torch.empty(M, ldq, dtype=dtype, device=dev)[:, :3 * E]
        attn_out = torch.empty(M, E, dtype=dtype, device=dev)
        proj = torch.empty(M, E, dtype=dtype, device=dev)
        mid = torch.empty(M, E, dtype=dtype, device=dev)
        FF = enc.layers[0].fc1.weight.shape[0]
        hbuf = torch.empty(M, FF, dtype=dtype, device=dev)
        # (residual + dropout in the GEMM epilogue: faster alone, slower inside the training step - csrc/gemm_pp2.hip)
        fused = dtype == torch.bfloat16 and os.environ.get('TELL_GEMM_RESIDUAL', '0') == '1'
        split_gelu = dtype == torch.bfloat16 and (M * FF) % 8 == 0 and os.environ.get('TELL_GELU_SPLIT', '0') == '1'
        for li, layer in enumerate(enc.layers):
            a = layer.self_attn
            wqkv, bqkv = self._qkv(a)
            ops.gemm(x, wqkv, out=qkv, bias=bqkv, bias_mode=1)
            # element (b,h,t,d) of q at qkv + (b*S + t)*3E + h*D + d  -> t-stride 3E, b-stride S*3E
            call('tell_attn_fwd', qkv, qkv[:, E:], qkv[:, 2 * E:], attn_out, None, pad_mask, None, None, B, H, S, S,
                 E // H, ldq, S * ldq, ldq, S * ldq, ldq, S * ldq, E, S * E, 0,
                 self.attention_dropout if tr else 0.0, rt.seed(), rt.next_salt() if tr else 0, dcode)
            l1 = layer.self_attn_layer_norm
            self._proj_residual_ln(attn_out, a.out_proj, x, l1, proj, mid, M, E, p_h, fused)
            if split_gelu:
                ops.gemm(mid, ops.weight(layer.fc1.weight), out=hbuf, bias=layer.fc1.bias.detach(), bias_mode=1)
                call('tell_gelu', hbuf, hbuf, M * FF, dcode)
            else:
                ops.gemm(mid, ops.weight(layer.fc1.weight), out=hbuf, bias=layer.fc1.bias.detach(), bias_mode=1, act=2)
            l2 = layer.final_layer_norm
            xn = stack[li + 1].view(M, E)
            self._proj_residual_ln(hbuf, layer.fc2, mid, l2, proj, xn, M, E, p_h, fused)
            x = xn
        return stack if return_all_hiddens else stack[-1]


def roberta_large(**kw):
    """random-initialised RoBERTa-large-shaped encoder (no checkpoint download offline);
    `load_state_dict` accepts fairseq's `model.decoder.sentence_encoder.*` names."""
    return RobertaEncoder(**kw)
