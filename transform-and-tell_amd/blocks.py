"""The residual blocks of a DynamicConv decoder layer (decoder_faces_objects.py:255-365) as single autograd nodes whose
forward and backward are explicit launch sequences, for the production case (training, bf16, post-LN, GLU + dynamic
convolution - every expt/ config):

  conv block  LN(X + dropout(linear2(DynamicConv(GLU(linear1(input_dropout(X)))))))            :256-266
  FFN block   LN(X + dropout(fc2(relu(fc1(X)))))                                               :357-364
  K|V of every (layer, context) pair in one grouped launch before the layer loop               multi_head.py:500-518

Composed from the per-op Functions of ops.py the same arithmetic costs ~60 launches per layer and step; what autograd
cannot know is that a block's input gradient is `residual gradient + branch gradient` (it adds them with a kernel of
its own), that the ReLU mask can ride in the epilogue of the GEMM that produces the masked gradient, and that the K/V
projections do not depend on the decoder state at all.  Here the last input-gradient GEMM of a block accumulates into
the residual gradient (or one pass applies the input-dropout mask and the sum), fc2's input gradient leaves its GEMM
already masked, the 16 context projections of a 4-layer / 4-context decoder are ONE launch, and the article's input
gradient (needed for the weigh_bert mix weights, transformer_faces_objects.py:355-364) is ONE long-K GEMM over all
layers instead of four GEMMs and three adds.

Parameter gradients never travel through autograd here either: weight gradients are queued for the pass's grouped
launches (ops.gemm_tn), LayerNorm gamma / beta and bias rows for its single column-sum launch (ops.finish_job)."""
import ctypes

import os

import torch
from torch.autograd import Function

from . import hip, ops
from . import runtime as rt

call = hip.call


def usable(layer, X):
    """The fused blocks cover: CUDA bf16 training with gradients, post-LN, GLU, the dynamic convolution with 64-wide
    heads - what every expt/ config runs.  Anything else keeps the per-op composition (models/decoders.py)."""
    from .modules import DynamicConv1dTBC
    return (X.is_cuda and X.dtype == torch.bfloat16 and rt.compute_dtype() == torch.bfloat16 and layer.training and
            torch.is_grad_enabled() and not layer.normalize_before and layer.glu and
            type(layer.conv) is DynamicConv1dTBC and
            layer.conv.input_size // layer.conv.num_heads == 64 and layer.embed_dim % 64 == 0 and
            getattr(layer.conv, 'weight_linear', None) is not None and layer.conv.weight_linear.bias is None and
            X.shape[-1] == layer.embed_dim and layer.conv_dim == layer.embed_dim)


def _ln_fwd(x2, r2, ln, p, salt, out=None):
    rows, C = x2.shape
    y = torch.empty(rows, C, dtype=x2.dtype, device=x2.device) if out is None else out
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    call('tell_layernorm_fwd', x2, x2.stride(0), r2, r2.stride(0), ln.weight.detach(), ln.bias.detach(), y, y.stride(0),
         mean, rstd, rows, C, float(ln.eps), float(p), rt.seed(), salt, hip.dt(x2))
    return y, mean, rstd


def _ln_bwd(dy2, x2, r2, ln, mean, rstd, p, salt):
    """-> (d branch [rows,C] with the dropout mask applied, d residual [rows,C]); gamma / beta gradients queued."""
    rows, C = x2.shape
    nb = hip.lib().tell_layernorm_bwd_blocks(rows)
    partial = torch.empty(nb, 2 * C, dtype=torch.float32, device=x2.device)
    dx = torch.empty_like(x2)
    dres = dx if p <= 0 else torch.empty_like(x2)
    defer = ops._FINISH['defer']
    gg, gb = ops.grad_buffer(ln.weight), ops.grad_buffer(ln.bias)
    call('tell_layernorm_bwd', dy2, dy2.stride(0), x2, x2.stride(0), r2, r2.stride(0), ln.weight.detach(), mean, rstd,
         dx, dx.stride(0), None if p <= 0 else dres, dres.stride(0), 0, None if defer else gg, None if defer else gb, 1,
         partial, rows, C, float(p), rt.seed(), salt, hip.dt(x2))
    if defer:
        ops.finish_job(partial, gg, C, gb)
    return dx, dres


# --------------------------------------------------------------------------------------------------------------------
class FFNBlockFn(Function):
    """LN(X + dropout(fc2(relu_dropout(relu(fc1(X))))))  - 3 launches forward, 3 backward (+ queued weight gradients)."""

    @staticmethod
    def forward(ctx, x, layer, salts):
        x2 = ops.as2dc(x)
        fc1, fc2 = layer.fc1, layer.fc2
        w1, n1 = ops.wn_weight(fc1.weight_g, fc1.weight_v)
        w2, n2 = ops.wn_weight(fc2.weight_g, fc2.weight_v)
        h = ops.gemm(x2, w1, bias=fc1.bias.detach(), bias_mode=1, act=1)
        p_r, p = layer.relu_dropout, layer.dropout
        hd = h
        if p_r > 0:
            hd = torch.empty_like(h)
            call('tell_dropout', h, hd, h.numel(), float(p_r), rt.seed(), salts[0], hip.dt(h))
        y2 = ops.gemm(hd, w2, bias=fc2.bias.detach(), bias_mode=1)
        out, mean, rstd = _ln_fwd(y2, x2, layer.final_layer_norm, p, salts[1])
        ctx.save_for_backward(x2, h, hd, y2, mean, rstd, w1, n1, w2, n2)
        ctx.layer, ctx.salts, ctx.shape = layer, salts, x.shape
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, h, hd, y2, mean, rstd, w1, n1, w2, n2 = ctx.saved_tensors
        layer, salts = ctx.layer, ctx.salts
        fc1, fc2 = layer.fc1, layer.fc2
        p_r, p = layer.relu_dropout, layer.dropout
        dy2, dres = _ln_bwd(ops.as2d(dout), y2, x2, layer.final_layer_norm, mean, rstd, p, salts[1])
        if dres is dy2:                                  # (no dropout: one tensor served both; the accumulation below
            dres = dy2.clone()                           #  would otherwise feed back into fc2's gradient)
        ops.wn_wgrad(dy2, hd, fc2.weight_g, fc2.weight_v, fc2.bias, n2)
        bt2 = lambda: ops.wn_weight_t(fc2.weight_g, fc2.weight_v)        # noqa: E731
        if p_r > 0:
            dhd = ops.gemm_nn(dy2, w2, b_t=bt2)
            if not dhd.is_contiguous():
                dhd = dhd.contiguous()
            dh = torch.empty_like(dhd)
            call('tell_dropout', dhd, dh, dhd.numel(), float(p_r), rt.seed(), salts[0], hip.dt(dhd))
            d = torch.empty_like(dh)
            call('tell_relu_bwd', dh, h, d, dh.numel(), hip.dt(dh))
            dh = d
        else:
            dh = torch.empty_like(h)
            ops.gemm_nn(dy2, w2, b_t=bt2, out=dh, act=3, aux=h)          # the ReLU mask rides in the GEMM's epilogue
        ops.wn_wgrad(dh, x2, fc1.weight_g, fc1.weight_v, fc1.bias, n1)
        ops.gemm_nn(dh, w1, b_t=lambda: ops.wn_weight_t(fc1.weight_g, fc1.weight_v), out=dres, accumulate=True)
        return dres.view(ctx.shape), None, None


def ffn_block(layer, X):
    salts = (rt.next_salt() if layer.relu_dropout > 0 else 0, rt.next_salt() if layer.dropout > 0 else 0)
    return FFNBlockFn.apply(X, layer, salts)


# --------------------------------------------------------------------------------------------------------------------
class ConvBlockFn(Function):
    """LN(X + dropout(linear2(DynamicConv(GLU(linear1(input_dropout(X)))))))."""

    @staticmethod
    def forward(ctx, x, layer, salts):
        T, B, E = x.shape
        x2 = ops.as2dc(x)
        conv, l1, l2 = layer.conv, layer.linear1, layer.linear2
        H, K = conv.num_heads, conv.kernel_size
        p_in, p_w, p = layer.input_dropout, conv.weight_dropout, layer.dropout
        x0 = x2
        if p_in > 0:
            x0 = torch.empty_like(x2)
            call('tell_dropout', x2, x0, x2.numel(), float(p_in), rt.seed(), salts[0], hip.dt(x2))
        w1, n1 = ops.wn_weight(l1.weight_g, l1.weight_v)
        w2, n2 = ops.wn_weight(l2.weight_g, l2.weight_v)
        h1 = ops.gemm(x0, w1, bias=l1.bias.detach(), bias_mode=1)                       # [rows, 2E]
        gl = torch.empty(x2.shape[0], E, dtype=x2.dtype, device=x2.device)
        wt = ops.weight(conv.weight_linear.weight)                                      # [H*K, E]
        c = torch.empty_like(gl)
        taps = torch.empty(T * B * H, K, dtype=torch.float32, device=x2.device)
        # GLU + tap logits + tap softmax + DropConnect + K-tap sum as ONE launch (csrc/dynconv.hip); it declines
        # (returns 1) shapes it does not take - fp32 parity mode, T > 32, other widths - and the three launches run
        fused = (x2.dtype == torch.bfloat16 and E == H * 64 and h1.is_contiguous() and wt.is_contiguous() and
                 hip.call_rc('tell_dynconv_block_fwd', h1, wt, gl, c, taps, T, B, H, K, float(p_w), rt.seed(),
                             salts[1]) == 0)
        if not fused:
            call('tell_glu_fwd', h1, gl, h1.shape[0], E, hip.dt(h1))
            logits = ops.gemm(gl, wt)
            call('tell_dynconv_fwd', gl, logits, c, taps, T, B, H, K, E // H, float(p_w), rt.seed(), salts[1], hip.dt(gl))
        y2 = ops.gemm(c, w2, bias=l2.bias.detach(), bias_mode=1)
        out, mean, rstd = _ln_fwd(y2, x2, layer.conv_layer_norm, p, salts[2])
        ctx.save_for_backward(x2, x0, h1, gl, taps, c, y2, mean, rstd, w1, n1, w2, n2, wt)
        ctx.layer, ctx.salts, ctx.shape = layer, salts, x.shape
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, x0, h1, gl, taps, c, y2, mean, rstd, w1, n1, w2, n2, wt = ctx.saved_tensors
        layer, salts = ctx.layer, ctx.salts
        T, B, E = ctx.shape
        conv, l1, l2 = layer.conv, layer.linear1, layer.linear2
        H, K = conv.num_heads, conv.kernel_size
        p_in, p_w, p = layer.input_dropout, conv.weight_dropout, layer.dropout
        dy2, dres = _ln_bwd(ops.as2d(dout), y2, x2, layer.conv_layer_norm, mean, rstd, p, salts[2])
        if dres is dy2:
            dres = dy2.clone()
        ops.wn_wgrad(dy2, c, l2.weight_g, l2.weight_v, l2.bias, n2)
        dc = ops.gemm_nn(dy2, w2, b_t=lambda: ops.wn_weight_t(l2.weight_g, l2.weight_v))
        if not dc.is_contiguous():
            dc = dc.contiguous()
        dgl = torch.empty_like(gl)
        dlogits = torch.empty(T * B, H * K, dtype=gl.dtype, device=gl.device)
        call('tell_dynconv_bwd', gl, dc, taps, dgl, 0, dlogits, T, B, H, K, E // H, float(p_w), rt.seed(), salts[1],
             hip.dt(gl))
        ops.linear_wgrad(dlogits, gl, conv.weight_linear.weight)
        ops.gemm_nn(dlogits, wt, b_t=lambda: ops.weight_t(conv.weight_linear.weight), out=dgl, accumulate=True)
        dh1 = torch.empty_like(h1)
        call('tell_glu_bwd', h1, dgl, dh1, h1.shape[0], E, hip.dt(h1))
        ops.wn_wgrad(dh1, x0, l1.weight_g, l1.weight_v, l1.bias, n1)
        bt1 = lambda: ops.wn_weight_t(l1.weight_g, l1.weight_v)         # noqa: E731
        if p_in > 0:
            dx0 = ops.gemm_nn(dh1, w1, b_t=bt1)
            if not dx0.is_contiguous():
                dx0 = dx0.contiguous()
            dx = torch.empty_like(dres)
            call('tell_dropout_add', dx0, dres, dx, dx0.numel(), float(p_in), rt.seed(), salts[0], hip.dt(dx0))
        else:
            dx = ops.gemm_nn(dh1, w1, b_t=bt1, out=dres, accumulate=True)
        return dx.view(ctx.shape), None, None


def conv_block(layer, X):
    salts = (rt.next_salt() if layer.input_dropout > 0 else 0, rt.next_salt() if layer.conv.weight_dropout > 0 else 0,
             rt.next_salt() if layer.dropout > 0 else 0)
    return ConvBlockFn.apply(X, layer, salts)


# --------------------------------------------------------------------------------------------------------------------
def _kv_weight(m):
    """(stacked [2E, kdim] working weight K rows over V rows, bias rows E:3E, fp32 gradient rows or None) of one
    MultiHeadAttention - the operands of ops.KVLinearFn."""
    E = m.embed_dim
    wk, rk = m._wrows(1)
    wv, rv = m._wrows(2)
    a, b = ops.weight(wk, rk), ops.weight(wv, rv)
    w = ops._stacked(a, b) if ops._adjacent(a, b) else \
        ops._cached(wk, ('kv', rk, id(wv), wv._version), lambda: torch.cat([a, b], 0))
    return w, m.in_proj_bias.detach()[E:3 * E], (wk, rk, wv, rv)


_KV_PP = os.environ.get('TELL_KV_PP', '1') != '0'          # A/B aid
_KV_PITCH_PAD = int(os.environ.get('TELL_KV_PITCH_PAD', '256'))      # elements; 0 = packed rows (A/B aid)
_KV_PITCH_PAD1 = int(os.environ.get('TELL_KV_PITCH_PAD1', '0'))    # the same for the single-layer buffers of the small contexts


class KVAllFn(Function):
    """The packed K|V projection [S,B,2E] of EVERY (layer, context) pair as one grouped launch: they depend on the
    static contexts and the layers' weights only, not on the decoder state, so none of them belongs inside the layer
    loop.  Inputs: the distinct context tensors ([S,B,kdim] views, batch-major storage used in place); outputs: one
    packed projection per job, in job order.

    The jobs of a context that needs an input gradient (the article: the weigh_bert mix weights are trained) write
    side by side into ONE [B*S, L*2E] buffer, and their attention backward passes are handed the matching slices of ONE
    gradient buffer (ops._DKV_TARGET): the input gradient is then dX = dKV_all [B*S, L*2E] . W_all [L*2E, kdim], a
    single long-K GEMM, instead of L GEMMs accumulated by L-1 adds."""

    @staticmethod
    def forward(ctx, jobs, n_ctx, *tensors):
        # jobs: [(attention module, context index)]; tensors: the n_ctx contexts, then the projections' parameters -
        # inputs only so that autograd sees the outputs depend on trainable tensors (their gradients are written
        # straight into the flat gradient buffer, never returned)
        ctxs = tensors[:n_ctx]
        ctx.n_params = len(tensors) - n_ctx
        srcs, bmaj = [], []
        for t in ctxs:
            bm = t.transpose(0, 1)
            use_bm = bm.is_contiguous() and not t.is_contiguous()   # batch-major storage (encoder outputs): used in place
            srcs.append(bm if use_bm else t.contiguous())           # [B,S,kdim] or [S,B,kdim]
            bmaj.append(use_bm)
        by_ctx = {}
        for j, (m, ci) in enumerate(jobs):
            by_ctx.setdefault(ci, []).append(j)
        outs, probs, meta, cat = [None] * len(jobs), [], [None] * len(jobs), {}

        def as_sbe(y2, ci, E2):                                     # [rows, 2E] in storage row order -> [S,B,2E] view
            src = srcs[ci]
            y3 = y2.view(src.shape[0], src.shape[1], E2)
            return y3.transpose(0, 1) if bmaj[ci] else y3
        for ci, js in by_ctx.items():
            s2 = srcs[ci].reshape(-1, srcs[ci].shape[-1])
            E2 = 2 * jobs[js[0]][0].embed_dim
            if ctxs[ci].requires_grad and len(js) > 1:              # side by side: one buffer, one gradient buffer
                # Row pitch n * 2E + 256 elements, NOT n * 2E: with 4 layers of 2E = 2048 bf16 a row is exactly 16 KB, and
                # an attention workgroup (b, h) walks 512 keys that are one row apart each - every 128-byte piece it
                # reads or writes then lands on the same few HBM channels.  MEASURED (tools/probes/attn_cold.py, cold
                # caches as in the step): article attention backward 103 -> 55 us, forward 43 -> 29 us with the pad.
                W = len(js) * E2
                buf = torch.empty(s2.shape[0], W + _KV_PITCH_PAD, dtype=s2.dtype, device=s2.device)[:, :W]
                dbuf = torch.empty(s2.shape[0], W + _KV_PITCH_PAD, dtype=s2.dtype, device=s2.device)[:, :W]
                cat[ci] = (buf, dbuf, js)
            for k, j in enumerate(js):
                w, bias, wmeta = _kv_weight(jobs[j][0])
                y = cat[ci][0][:, k * E2:(k + 1) * E2] if ci in cat else \
                    torch.empty(s2.shape[0], E2 + _KV_PITCH_PAD1, dtype=s2.dtype, device=s2.device)[:, :E2]
                probs.append(dict(a=s2, b=w, out=y, form='nt', bias=bias))
                outs[j] = as_sbe(y, ci, E2)
                meta[j] = (w, wmeta, s2)
                if ci in cat:
                    ops._DKV_TARGET[outs[j].data_ptr()] = as_sbe(cat[ci][1][:, k * E2:(k + 1) * E2], ci, E2)
        # the article's projections (B*S = 16384 rows at B = 32) are whole rounds of 256x256 tiles: the ping-pong GEMM
        # runs them at ~800 TFLOP/s, the grouped 128x128 kernel at ~520; everything small stays one grouped launch
        big = [q for q in probs if _KV_PP and q['a'].shape[0] >= 8192 and q['a'].shape[0] % 256 == 0 and
               q['b'].shape[0] % 256 == 0]
        for q in big:
            ops.gemm(q['a'], q['b'], out=q['out'], bias=q['bias'], bias_mode=1 if q['bias'] is not None else 0)
        rest = [q for q in probs if not any(q is b_ for b_ in big)]
        if rest:
            ops.gemm_grouped(rest)
        ctx.jobs, ctx.meta, ctx.cat, ctx.by_ctx, ctx.bmaj = jobs, meta, cat, by_ctx, bmaj
        ctx.need_dx = [t.requires_grad for t in ctxs]
        ctx.src_shapes = [tuple(t.shape) for t in srcs]
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dkvs):
        jobs, meta, cat, by_ctx, bmaj = ctx.jobs, ctx.meta, ctx.cat, ctx.by_ctx, ctx.bmaj
        dctx = [None] * len(ctx.need_dx)
        for ci, js in by_ctx.items():
            d2s = []
            for k, j in enumerate(js):
                d = dkvs[j]
                m = jobs[j][0]
                E = m.embed_dim
                w, (wk, rk, wv, rv), s2 = meta[j]
                if d is None:
                    d2s.append(None)
                    continue
                d3 = d.transpose(0, 1) if bmaj[ci] else d           # storage row order of the projection (= of s2)
                if d3.stride(2) != 1 or d3.stride(0) != d3.shape[1] * d3.stride(1):
                    d3 = d3.contiguous()
                d2 = d3.reshape(-1, 2 * E)                          # a view: the two row dims merge
                d2s.append(d2)
                b_param = m.in_proj_bias
                gb = ops.grad_buffer(b_param)[E:3 * E] if b_param.requires_grad else None
                if wk.requires_grad:
                    (gk, acc_k), (gv, acc_v) = ops.wgrad_target(wk, rk), ops.wgrad_target(wv, rv)
                    ops.kv_wgrad(d2, s2, gk, acc_k, gv, acc_v, gb, E)
                elif gb is not None:
                    ops.colsum_into(d2, gb)
            if not ctx.need_dx[ci]:
                continue
            dx = None
            if ci in cat and all(d is not None for d in d2s):
                buf, dbuf, _ = cat[ci]
                n = len(js)
                E2 = dbuf.shape[1] // n
                whole = all(d2s[k].data_ptr() == dbuf[:, k * E2:(k + 1) * E2].data_ptr() and
                            d2s[k].stride(0) == dbuf.stride(0) for k in range(n))
                if whole:
                    # dX = dKV_all . W_all as an NT product against W_all^T [kdim, L*2E] (one launch builds it)
                    ws = [meta[j][0] for j in js]
                    wt = torch.empty(ws[0].shape[1], dbuf.shape[1], dtype=dbuf.dtype, device=dbuf.device)
                    call('tell_transpose_multi', n, ops._ptr_array(ws), (ctypes.c_long * n)(*[w.stride(0) for w in ws]),
                         ops._ptr_array([wt[:, k * E2:(k + 1) * E2] for k in range(n)]),
                         (ctypes.c_long * n)(*[wt.stride(0)] * n), ops._int_array([w.shape[0] for w in ws]),
                         ops._int_array([w.shape[1] for w in ws]))
                    dx = ops.gemm(dbuf, wt)
            if dx is None:
                for k, j in enumerate(js):
                    if d2s[k] is None:
                        continue
                    w = meta[j][0]
                    if dx is None:
                        dx = ops.gemm_nn(d2s[k], w, b_t=lambda w=w: ops.transpose(w)[0])
                        if not dx.is_contiguous():
                            dx = dx.contiguous()
                    else:
                        ops.gemm_nn(d2s[k], w, b_t=lambda w=w: ops.transpose(w)[0], out=dx, accumulate=True)
            if dx is not None:
                d3 = dx.view(ctx.src_shapes[ci])
                dctx[ci] = d3.transpose(0, 1) if bmaj[ci] else d3
        return (None, None) + tuple(dctx) + (None,) * ctx.n_params


def kv_project_all(layers, names, contexts):
    """-> [ {name: packed K|V [S,B,2E]} per layer ] for the non-empty contexts (empty ones are handled by the attention
    module itself: multi_head.py:349-374)."""
    ctxs, idx, jobs = [], {}, []
    per_layer = [dict() for _ in layers]
    for name in names:
        t = contexts.get(name)
        if not torch.is_tensor(t) or t.dim() != 3 or t.shape[0] == 0 or t.shape[2] == 0 or t.shape[2] % 64 != 0:
            continue
        idx[name] = len(ctxs)
        ctxs.append(t)
    for name, ci in idx.items():
        for li, layer in enumerate(layers):
            jobs.append((layer.context_attns[name], ci, li, name))
    if not jobs:
        return per_layer
    ops._DKV_TARGET.clear()                                  # (targets of a previous, unfinished pass)
    params = []
    for m, _, _, _ in jobs:
        wk, _ = m._wrows(1)
        wv, _ = m._wrows(2)
        params += [wk, m.in_proj_bias] if wv is wk else [wk, wv, m.in_proj_bias]
    outs = KVAllFn.apply([(m, ci) for m, ci, _, _ in jobs], len(ctxs), *ctxs, *params)
    for (m, ci, li, name), y in zip(jobs, outs):
        per_layer[li][name] = y
    return per_layer


def prepare_transposes(layers, rows):
    """W^T of the weight-normalised weights whose input-gradient GEMM takes the NT kernel (>= 256 tiles of 128x128:
    fc2 and context_fc at rows = 1024), for all layers in ONE launch; fills the cache ops.wn_weight_t reads."""
    import weakref
    todo = []
    for layer in layers:
        for lin in (layer.fc2, layer.context_fc, layer.fc1, layer.linear1, layer.linear2):
            g, v = lin.weight_g, lin.weight_v
            n_out, n_in = v.shape
            big = ((rows + 127) // 128) * ((n_in + 127) // 128) >= 256 and n_out % 64 == 0
            key = ('wn_t', g._version, g.data_ptr())
            if big and v.is_cuda and ops._fresh(v, key) is None:
                todo.append((g, v, key))
    if not todo:
        return
    ws = [ops.wn_weight(g, v)[0] for g, v, _ in todo]
    n = len(todo)
    dsts = [torch.empty(w.shape[1], w.shape[0], dtype=w.dtype, device=w.device) for w in ws]
    call('tell_transpose_multi', n, ops._ptr_array(ws), (ctypes.c_long * n)(*[w.stride(0) for w in ws]),
         ops._ptr_array(dsts), (ctypes.c_long * n)(*[d.stride(0) for d in dsts]), ops._int_array([w.shape[0] for w in ws]),
         ops._int_array([w.shape[1] for w in ws]))
    for (g, v, key), d in zip(todo, dsts):
        ops._wcache[(id(v), key)] = (ops._stamp(v), d, weakref.ref(v))
