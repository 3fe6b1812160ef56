"""Pure-function CPU restatements (fp32 torch) of the hot-path arithmetic.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Citations are relative to
/root/reference.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # torch.nn.LayerNorm default, used everywhere in the reference


# --------------------------------------------------------------------------- #
# weight-normalised linear  (tell/modules/linear.py:8-33)
# --------------------------------------------------------------------------- #
def weight_norm_weight(weight_g, weight_v):
    """W[r,:] = g[r] * v[r,:] / ||v[r,:]||_2   (torch weight_norm, dim=0;
    tell/modules/linear.py:33)."""
    norms = weight_v.norm(dim=1, keepdim=True)
    return weight_v * (weight_g / norms)


def gehring_linear(x, weight_g, weight_v, bias=None):
    return F.linear(x, weight_norm_weight(weight_g, weight_v), bias)


# --------------------------------------------------------------------------- #
# dynamic convolution  (tell/modules/convolutions/dynamic.py:285-336)
# --------------------------------------------------------------------------- #
def dynamic_conv_taps(tap_logits, n_heads, kernel_size, drop_mask=None, p=0.0):
    """softmax over ALL K taps (dynamic.py:302-304) then DropConnect (:305).
    tap_logits: [T,B,H*K] -> [T,B,H,K]."""
    T, B, _ = tap_logits.shape
    w = torch.softmax(tap_logits.view(T, B, n_heads, kernel_size), dim=-1)
    if drop_mask is not None:
        w = w * drop_mask / (1.0 - p)
    return w


def dynamic_conv_apply(x, taps):
    """out[t,b,h,:] = sum_k taps[t,b,h,k] * x[t-(K-1)+k, b, h, :], x[<0] = 0.

    Equivalent to the reference's band-matrix bmm (dynamic.py:318-335) including
    its K>T narrowing (:324-326), which only drops taps that would have hit the
    zero padding anyway."""
    T, B, C = x.shape
    H, K = taps.shape[2], taps.shape[3]
    R = C // H
    xh = x.view(T, B, H, R)
    xp = torch.cat([x.new_zeros(K - 1, B, H, R), xh], dim=0)  # [T+K-1,B,H,R]
    out = x.new_zeros(T, B, H, R)
    for k in range(K):
        out = out + taps[..., k:k + 1] * xp[k:k + T]
    return out.view(T, B, C)


# --------------------------------------------------------------------------- #
# multi-head cross attention (tell/modules/attention/multi_head.py:288-486,
# generic path taken when static_kv=True, incremental_state=None)
# --------------------------------------------------------------------------- #
def cross_attention(x, ctx, key_padding_mask, wq, wk, wv, in_bias, bias_k, bias_v,
                    wo, bo, n_heads, need_weights=False, drop_mask=None, p=0.0):
    """x [T,B,E]; ctx [S,B,kdim] (kdim may be 0: empty-context branch,
    multi_head.py:349-374); key_padding_mask [B,S] bool or None."""
    T, B, E = x.shape
    hd = E // n_heads
    q = F.linear(x, wq, in_bias[:E]) * hd ** -0.5                      # :348-353
    if ctx.shape[2] > 0:
        k = F.linear(ctx, wk, in_bias[E:2 * E])                        # :500-509
        v = F.linear(ctx, wv, in_bias[2 * E:])                         # :511-518
        k = torch.cat([k, bias_k.expand(1, B, E)], dim=0)              # :355-364
        v = torch.cat([v, bias_v.expand(1, B, E)], dim=0)
        if key_padding_mask is not None:
            key_padding_mask = torch.cat(
                [key_padding_mask, key_padding_mask.new_zeros(B, 1)], dim=1)
    else:
        k = bias_k.expand(1, B, E)
        v = bias_v.expand(1, B, E)
        if key_padding_mask is not None:
            key_padding_mask = key_padding_mask.new_zeros(B, 1)        # :372-374
    S1 = k.shape[0]
    q = q.reshape(T, B * n_heads, hd).transpose(0, 1)                   # :376-380
    k = k.reshape(S1, B * n_heads, hd).transpose(0, 1)
    v = v.reshape(S1, B * n_heads, hd).transpose(0, 1)
    k = torch.cat([k, k.new_zeros(B * n_heads, 1, hd)], dim=1)          # :416-421
    v = torch.cat([v, v.new_zeros(B * n_heads, 1, hd)], dim=1)
    if key_padding_mask is not None:
        key_padding_mask = torch.cat(
            [key_padding_mask, key_padding_mask.new_zeros(B, 1)], dim=1)  # :425-427
    scores = torch.bmm(q, k.transpose(1, 2))                            # :429
    if key_padding_mask is not None:
        scores = scores.view(B, n_heads, T, S1 + 1).masked_fill(
            key_padding_mask[:, None, None, :], float('-inf')).view(B * n_heads, T, S1 + 1)
    probs = torch.softmax(scores.float(), dim=-1).type_as(scores)       # :460-462
    pd = probs
    if drop_mask is not None:
        pd = probs * drop_mask / (1.0 - p)                              # :463
    o = torch.bmm(pd, v).transpose(0, 1).reshape(T, B, E)               # :466-475
    o = F.linear(o, wo, bo)                                             # :476
    w = None
    if need_weights:
        w = pd.view(B, n_heads, T, S1 + 1).sum(dim=1) / n_heads         # :478-482
    return o, w


# --------------------------------------------------------------------------- #
# embeddings (tell/modules/token_embedders/adaptive.py:61-76,
#             tell/modules/token_embedders/positional.py:126-268)
# --------------------------------------------------------------------------- #
def adaptive_embed(ids, cutoffs, emb_weights, proj_weights, scale):
    """ids: int64 [...]; band i covers [cutoffs[i-1], cutoffs[i])."""
    out = emb_weights[0].new_zeros(ids.shape + (proj_weights[0].shape[0],))
    lo = 0
    for i, hi in enumerate(cutoffs):
        m = (ids >= lo) & (ids < hi)
        if m.any():
            out[m] = F.linear(emb_weights[i][ids[m] - lo], proj_weights[i]).to(out.dtype)   # (.to: autocast runs)
        lo = hi
    return out * scale


def sinusoid_table(n_rows, dim, padding_idx):
    """positional.py:126-165: [sin | cos] halves (not interleaved), geometric
    timescales 1..1e4, row `padding_idx` zeroed."""
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


def make_positions(ids, padding_idx, left_pad=False):
    """positional.py:231-268.  Non-pad symbol in column j -> padding_idx+1+j
    (minus the row's pad count when left-padded); pad -> padding_idx."""
    n = ids.shape[1]
    pos = torch.arange(padding_idx + 1, padding_idx + 1 + n, dtype=ids.dtype)[None, :].expand_as(ids)
    mask = ids.ne(padding_idx)
    if left_pad:
        pos = pos - (n - mask.long().sum(dim=1, keepdim=True))
    return torch.where(mask, pos, torch.full_like(ids, padding_idx))


# --------------------------------------------------------------------------- #
# adaptive softmax + loss (tell/modules/softmax.py:144-222,
#                          tell/modules/criteria/adaptive_loss.py:27-73)
# --------------------------------------------------------------------------- #
def adaptive_log_probs(x, cutoffs, head_word_w, head_class_w, tail_proj_ws, tail_emb_ws):
    """Full-vocabulary log-probs, softmax.py:193-222.  x: [N,E]."""
    head = torch.log_softmax(F.linear(x, torch.cat([head_word_w, head_class_w], dim=0)), dim=1)
    pieces = [head[:, :cutoffs[0]]]
    for i in range(len(tail_proj_ws)):
        t = torch.log_softmax(F.linear(F.linear(x, tail_proj_ws[i]), tail_emb_ws[i]), dim=1)
        pieces.append(t + head[:, cutoffs[0] + i, None])
    return torch.cat(pieces, dim=1)


def adaptive_loss_sum(x, target, cutoffs, head_word_w, head_class_w, tail_proj_ws,
                      tail_emb_ws, padding_idx=1):
    """Summed (nats) adaptive-softmax cross entropy + sample_size.

    softmax.py:144-191 remaps targets of band i>=1 to head class cutoff[0]+i-1
    and computes tail logits only for the rows of that band;
    adaptive_loss.py:55-60 sums `cross_entropy(..., ignore_index=padding_idx,
    reduction='sum')` over clusters.  Quirk reproduced: ignore_index is applied
    to the *tail-local* index too, so global targets cutoff[i]+1 (5001, 20001)
    contribute a head term but no tail term."""
    x = x.reshape(-1, x.shape[-1])
    target = target.reshape(-1)
    head_t = target.clone()
    loss = x.new_zeros(())
    for i in range(len(tail_proj_ws)):
        band = (target >= cutoffs[i]) & (target < cutoffs[i + 1])
        head_t[band] = cutoffs[0] + i
        if band.any():
            rows = band.nonzero().squeeze(1)
            logits = F.linear(F.linear(x[rows], tail_proj_ws[i]), tail_emb_ws[i])
            loss = loss + F.cross_entropy(logits, target[rows] - cutoffs[i],
                                          ignore_index=padding_idx, reduction='sum')
    head_logits = F.linear(x, torch.cat([head_word_w, head_class_w], dim=0))
    loss = loss + F.cross_entropy(head_logits, head_t, ignore_index=padding_idx, reduction='sum')
    sample_size = int((target != padding_idx).sum())          # adaptive_loss.py:62-65
    return loss, sample_size
