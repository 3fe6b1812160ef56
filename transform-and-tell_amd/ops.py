"""Functional ops of the MI355X path: thin `torch.autograd.Function`s whose forward
and backward are sequences of C-ABI kernel launches (hip.call).  torch supplies
device memory, the current stream and the autograd graph; no arithmetic of the
hot path is done by ATen here.

Gradient convention: parameters are fp32 masters.  Backward kernels accumulate
weight gradients *directly* into the parameter's fp32 gradient buffer
(`grad_buffer(p)`, which is `p.grad`), the way fused wgrad accumulation works in
large-model trainers; the Functions therefore return None for parameters.
"""
import ctypes
import math
import os
import weakref

import torch
from torch.autograd import Function

from . import hip, prof, streams
from . import runtime as rt

call = hip.call


def _vec(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def _round_up(n, m):
    return (n + m - 1) // m * m


def as2d(x):
    """[..., C] -> [rows, C] with unit inner stride (row stride may exceed C)."""
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.stride(-1) == 1 else x2.contiguous()


def as2dc(x):
    return x.reshape(-1, x.shape[-1]).contiguous()


_GRAD_STORE = {'on': False, 'observe': None}          # (see 'Gradient stores' below)


def grad_buffer(p, _dense_writer=False):
    """fp32 gradient accumulator of a parameter (== p.grad).  While the trainer observes a step (see _GRAD_STORE) every
    caller but the dense weight-gradient sites counts as an ACCUMULATING writer: such a parameter keeps zero + accumulate."""
    if not _dense_writer and _GRAD_STORE['observe'] is not None:
        _GRAD_STORE['observe'].setdefault(id(p), (p, []))[1].append(None)
    if p.grad is None:
        g = getattr(p, '_tell_grad', None)
        if g is None or g.shape != p.shape or g.device != p.device:
            g = torch.zeros_like(p)
        else:
            g.zero_()
        p._tell_grad = g
        p.grad = g
    return p.grad


# --------------------------------------------------------------------------- #
# raw kernels
# --------------------------------------------------------------------------- #
_SPLITK = os.environ.get('TELL_GEMM_SKINNY_SPLITK', '1') != '0'          # A/B aid
_COUNT_LIMIT = os.environ.get('TELL_COUNT_LIMIT', '1') != '0'            # A/B aid: 0 = count-limited buffers are worked on whole


# Gradient stores (round 4).  The flat gradient buffer starts every step zeroed and the backward kernels accumulate into
# it - for a weight whose gradient is ONE dense product per step that is 4 B / parameter of zeros read back by the product
# and 4 B / parameter of zeros written by the optimizer.  The trainer OBSERVES its first eager step (who writes which
# rows of which parameter through wgrad_target / note_grad_write); parameters written exactly once, whole, get
# `_tell_grad_store`: from then on their product stores (accumulate = 0) and BertAdam leaves their gradient alone
# (tell_bertadam_step2 keep_grad).  Everything else - tied tables, biases, LayerNorm parameters, scatter-added
# embeddings - keeps the zero + accumulate convention.


def grad_store_on():
    return _GRAD_STORE['on']


def grad_store_mode(on):
    _GRAD_STORE['on'] = bool(on)


def grad_store_observe(start):
    """start=True: begin recording gradient writes; start=False: stop and return {id(p): (p, [(r0, r1), ...])}."""
    if start:
        _GRAD_STORE['observe'] = {}
        return None
    seen, _GRAD_STORE['observe'] = _GRAD_STORE['observe'], None
    return seen or {}


def note_grad_write(p, rows=None):
    """A backward product is about to write rows [r0, r1) (default: all) of p's gradient, densely."""
    obs = _GRAD_STORE['observe']
    if obs is not None:
        n = p.shape[0]
        obs.setdefault(id(p), (p, []))[1].append((0, n) if rows is None else (int(rows[0]), int(rows[1])))


def grad_stored(p):
    """True: p's gradient is stored by its single producer this step (no accumulate, not zeroed by the optimizer)."""
    return _GRAD_STORE['on'] and getattr(p, '_tell_grad_store', False)


def wgrad_target(p, rows=None):
    """-> (rows of p's flat gradient as [rows, fan_in], accumulate flag) for a dense weight-gradient product."""
    gw = grad_buffer(p, True)
    gw2 = gw.view(gw.shape[0], -1)
    note_grad_write(p, rows)
    return (gw2 if rows is None else gw2[rows[0]:rows[1]]), not grad_stored(p)


def gemm(a, b, out=None, out_dtype=None, bias=None, bias_mode=0, act=0, aux=None, alpha=1.0,
         accumulate=False, m_dev=None):
    """out[M,N] = act((a[M,K] @ b[N,K]^T + bias) * alpha) (+ out).  K is zero padded to a
    16-byte multiple when needed (only reduced-size test shapes take that path)."""
    v = _vec(a.dtype)
    Kp = _round_up(max(a.shape[1], b.shape[1]), v)
    assert a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype and \
        _round_up(a.shape[1], v) == _round_up(b.shape[1], v), (a.shape, b.shape, a.dtype, b.dtype)
    M = a.shape[0]
    N = b.shape[0]

    def fix(t):
        if t.stride(1) != 1 or t.stride(0) % v or t.shape[1] != Kp or t.data_ptr() % 16:
            tp = torch.zeros(t.shape[0], Kp, dtype=t.dtype, device=t.device)
            tp[:, :t.shape[1]] = t
            return tp
        return t
    a, b = fix(a), fix(b)
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
    assert out.stride(1) == 1
    K = a.shape[1]
    if (_SPLITK and M <= 128 and K >= 2048 and K % 512 == 0 and N % 64 == 0 and a.dtype == torch.bfloat16 and
            bias_mode in (0, 1) and act in (0, 1, 2) and aux is None and not accumulate and m_dev is None and
            out.stride(0) % 4 == 0):
        # skinny and long (the decode step's context_fc / fc2: <= 128 rows x K = 4096): K slices as one grouped launch
        # of fp32 partial tiles + one fold-and-epilogue launch, instead of 16-32 workgroups walking 64 K tiles each
        splits = K // 512
        partial = torch.empty(splits, M, N, dtype=torch.float32, device=a.device)
        ks = K // splits
        gemm_grouped([dict(a=a[:, i * ks:(i + 1) * ks], b=b[:, i * ks:(i + 1) * ks], out=partial[i], form='nt')
                      for i in range(splits)])
        call('tell_splitk_reduce', partial, splits, partial.stride(0), M, N, bias if bias_mode == 1 else None, act,
             float(alpha), out, out.stride(0), hip.dt(out))
        return out
    args = (a, a.stride(0), b, b.stride(0), out, out.stride(0), M, N, a.shape[1], hip.dt(a), hip.dt(out), bias, bias_mode,
            act, aux, float(alpha), int(accumulate), m_dev)
    rec = slot = None
    if prof.enabled() and m_dev is None and 2.0 * M * N * a.shape[1] >= prof.MIN_WORK:
        kname = hip.query('tell_gemm_nt_plan', *args)          # the library names the kernel it is about to launch
        if torch.cuda.is_current_stream_capturing():           # no events inside a captured graph: device timestamps
            slot = prof.graph_begin(kname, 2.0 * M * N * a.shape[1])
        elif prof.sampled(kname):
            rec = prof.begin(kname, 2.0 * M * N * a.shape[1])
    call('tell_gemm_nt', *args)
    if rec is not None:
        prof.end(rec)
    if slot is not None:
        prof.graph_end(slot)
    return out


def _kmajor_ok(t):
    """bf16 [K, X] operand usable in place by the K-major GEMM forms (tell_gemm_bf16)."""
    return (t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and
            t.data_ptr() % 16 == 0)


def gemm_tn(a_km, b_kn, out=None, out_dtype=None, accumulate=False, alpha=1.0, m_dev=None, asum=None,
            asum_scale=1.0, k_dev=None):
    """out[M,N] = alpha * sum_k a_km[k,m] * b_kn[k,n] (+ out): both operands K-major, as the forward pass left
    them (wgrad: a_km = dY [T,N_out], b_kn = X [T,K_in]).  bf16 operands go to the LDS-transpose-read kernel;
    anything else (fp32 parity mode, unaligned strides) takes explicit transposes + the NT kernel.
    asum: optional fp32 [M] accumulator, asum += asum_scale * column sums of a_km (the bias gradient).
    k_dev: optional device int32 - only the first *k_dev rows of the two operands carry data (the rest are zeros: the
    count-limited buffers of the adaptive softmax); the K-major kernels stop reading there."""
    K, M = a_km.shape
    N = b_kn.shape[1]
    if not _COUNT_LIMIT:
        k_dev = None
    if _kmajor_ok(a_km) and _kmajor_ok(b_kn) and m_dev is None:
        if out is None:
            out = torch.empty(M, N, dtype=out_dtype or a_km.dtype, device=a_km.device)
        assert out.stride(1) == 1
        if _WGRAD_GROUP['defer'] and K > 0:
            # queued: the pass's weight-gradient GEMMs run together once it is over (wgrad_group_flush)
            _WGRAD_GROUP['items'].append((a_km, b_kn, out, float(alpha), int(accumulate), asum, float(asum_scale), k_dev))
            return out
        if k_dev is not None:
            gemm_grouped([dict(a=a_km, b=b_kn, out=out, form='tn', alpha=alpha, accumulate=accumulate, asum=asum,
                               asum_scale=asum_scale, lim_dev=k_dev)])
            return out
        call('tell_gemm_bf16', a_km, a_km.stride(0), 1, b_kn, b_kn.stride(0), 1, out, out.stride(0), M, N, K,
             hip.dt(out), None, 0, 0, None, float(alpha), int(accumulate), None, asum, float(asum_scale))
        return out
    if asum is not None:
        colsum_into(a_km, asum, scale=asum_scale)
    at, _ = transpose(a_km)
    bt, _ = transpose(b_kn)
    return gemm(at, bt, out=out, out_dtype=out_dtype, accumulate=accumulate, alpha=alpha, m_dev=m_dev)


class _GemmProblem(ctypes.Structure):
    """tell_gemm_problem (include/tell_hip.h)."""
    _fields_ = [('A', ctypes.c_void_p), ('lda', ctypes.c_long), ('B', ctypes.c_void_p), ('ldb', ctypes.c_long),
                ('C', ctypes.c_void_p), ('ldc', ctypes.c_long), ('M', ctypes.c_int), ('N', ctypes.c_int),
                ('K', ctypes.c_int), ('trans_a', ctypes.c_int), ('trans_b', ctypes.c_int),
                ('out_dtype', ctypes.c_int), ('accumulate', ctypes.c_int), ('alpha', ctypes.c_float),
                ('bias', ctypes.c_void_p), ('bias_mode', ctypes.c_int), ('act', ctypes.c_int),
                ('asum', ctypes.c_void_p), ('asum_scale', ctypes.c_float), ('lim_dev', ctypes.c_void_p)]


def gemm_grouped(problems):
    """Independent bf16 products in a few launches (tell_gemm_grouped).  problems: dicts with a, b, out (tensors),
    form ('nt': a [M,K], b [N,K]; 'nn': b stored [K,N]; 'tn': a stored [K,M] as well) and optionally alpha,
    accumulate, bias (fp32 [N], 'nt' only), act, asum, asum_scale, lim_dev (device int32 tensor: the number of valid rows
    of the batch dimension - output rows of 'nt' / 'nn', reduction rows of 'tn')."""
    if not problems:
        return
    arr = (_GemmProblem * len(problems))()
    for q, pr in zip(arr, problems):
        a, b, out, form = pr['a'], pr['b'], pr['out'], pr['form']
        assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.stride(1) == 1 and b.stride(1) == 1 and \
            out.stride(1) == 1
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), \
            out.stride(0)
        q.trans_a, q.trans_b = int(form == 'tn'), int(form in ('tn', 'nn'))
        q.M = a.shape[1] if form == 'tn' else a.shape[0]
        q.N = b.shape[0] if form == 'nt' else b.shape[1]
        q.K = b.shape[1] if form == 'nt' else b.shape[0]            # ('nn': dY may carry zero padding beyond K)
        q.out_dtype, q.accumulate, q.alpha = hip.dt(out), int(pr.get('accumulate', False)), float(pr.get('alpha', 1.0))
        bias = pr.get('bias')
        q.bias, q.bias_mode, q.act = (bias.data_ptr() if bias is not None else None), int(bias is not None), \
            int(pr.get('act', 0))
        asum = pr.get('asum')
        q.asum, q.asum_scale = (asum.data_ptr() if asum is not None else None), float(pr.get('asum_scale', 1.0))
        lim = pr.get('lim_dev')                  # device int32: valid rows of the batch dimension (M of nt / nn, K of tn)
        q.lim_dev = lim.data_ptr() if lim is not None else None
    call('tell_gemm_grouped', len(problems), arr)


_WGRAD_GROUP = {'defer': False, 'items': [], 'enabled': os.environ.get('TELL_WGRAD_GROUP', '1') != '0'}


def wgrad_group_defer(on):
    """While on, K-major weight-gradient GEMMs (gemm_tn) are queued instead of launched; wgrad_group_flush() runs the
    queue as a few grouped launches (tell_gemm_tn_grouped).  The trainer turns it on around backward when no gradient
    bucket has to leave before the pass is over."""
    _WGRAD_GROUP['defer'] = bool(on) and _WGRAD_GROUP['enabled']


def wgrad_group_drop():
    _WGRAD_GROUP['items'] = []


def wgrad_group_flush():
    items, _WGRAD_GROUP['items'] = _WGRAD_GROUP['items'], []
    # two products into overlapping outputs (tied projections accumulate into one gradient buffer) must not share a
    # launch: the later one waits for the next round
    while items:
        now, later, spans = [], [], []
        for it in items:
            out = it[2]
            lo = out.data_ptr()
            hi = lo + ((out.shape[0] - 1) * out.stride(0) + out.shape[1]) * out.element_size()
            if any(lo < h and l < hi for l, h in spans):
                later.append(it)
            else:
                spans.append((lo, hi))
                now.append(it)
        _launch_wgrad_group(now)
        items = later


def _launch_wgrad_group(items):
    gemm_grouped([dict(a=a, b=b, out=out, form='tn', alpha=alpha, accumulate=acc, asum=asum, asum_scale=asum_scale, lim_dev=kd)
                  for a, b, out, alpha, acc, asum, asum_scale, kd in items])


def gemm_nn(a, b_kn, b_t=None, out=None, out_dtype=None, alpha=1.0, act=0, aux=None, m_dev=None, accumulate=False,
            zero_rows=False):
    """out[M,N] = act(alpha * a[M,K] @ b_kn[K,N]) (+ out) (dgrad: a = dY, b_kn = the weight as stored [N_out, K_in]).
    `b_t`: callable returning b_kn^T [N, K] for the fallback path.  m_dev: device row count - rows past it are left
    untouched; zero_rows=True: the caller guarantees that those rows of `a` are zero and may be written (as zeros)."""
    M = a.shape[0]
    K, N = b_kn.shape                  # `a` may carry zero padding columns beyond K (never garbage: 0 * NaN)
    assert a.shape[1] >= K
    big = ((M + 127) // 128) * ((N + 127) // 128) >= 256 and K % 64 == 0     # direct-to-LDS NT kernel territory
    if (_SPLITK and K >= 8192 and N <= 1024 and M <= 1024 and N % 4 == 0 and _kmajor_ok(b_kn) and a.dtype == torch.bfloat16 and
            a.stride(1) == 1 and a.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and a.shape[1] >= _round_up(K, 8) and
            act == 0 and aux is None and not accumulate and (out is None or out.stride(0) % 4 == 0) and
            (m_dev is None or zero_rows)):
        # few output columns from a very long reduction (the adaptive-softmax tails' dh = dlogits . W: [1024, 64] from
        # K = 30265): 16-64 output tiles walking hundreds of K tiles each (162 us).  K slices as one grouped launch
        # of fp32 partial tiles + the fold.  (Rows past *m_dev are computed too - zero_rows: zeros in, zeros out.)
        # With adaptive_softmax_factor 1 (the bench configuration) the tails keep the model width: [1024, 1024] from
        # K = 15000 / 30265 is 256 tiles of 64x64 walking 235 / 473 K tiles (97 / 158 us); wider slices keep the fp32
        # partials at 8-16 MB.
        if out is None:
            out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
        ks = 2048 if N <= 256 else 8192
        splits = (K + ks - 1) // ks
        partial = torch.empty(splits, M, N, dtype=torch.float32, device=a.device)
        # (m_dev: workgroups of row tiles past the count leave at once, the fold skips those rows - they stay the zeros the
        #  caller's buffer holds.  With real captions the tails see a few dozen rows of the 1024-row capacity.)
        gemm_grouped([dict(a=a[:, i * ks:min(_round_up(K, 8), (i + 1) * ks)], b=b_kn[i * ks:min(K, (i + 1) * ks)],
                           out=partial[i], form='nn', lim_dev=m_dev if _COUNT_LIMIT else None) for i in range(splits)])
        call('tell_splitk_reduce2', partial, splits, partial.stride(0), M, N, None, 0, float(alpha), out, out.stride(0),
             hip.dt(out), m_dev if _COUNT_LIMIT else None)
        return out
    if (_kmajor_ok(b_kn) and a.dtype == torch.bfloat16 and a.stride(1) == 1 and a.stride(0) % 8 == 0 and
            (K % 8 == 0 or a.shape[1] >= _round_up(K, 8)) and a.data_ptr() % 16 == 0 and
            not (big and b_t is not None)):
        if out is None:
            out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
        call('tell_gemm_bf16', a, a.stride(0), 0, b_kn, b_kn.stride(0), 1, out, out.stride(0), M, N, K,
             hip.dt(out), None, 0, act, aux, float(alpha), int(accumulate), m_dev, None, 0.0)
        return out
    bt = b_t() if b_t is not None else transpose(b_kn)[0]
    if bt.shape[1] != a.shape[1]:                       # bring both to one (zero padded) K width
        Kp = max(bt.shape[1], a.shape[1])
        if bt.shape[1] != Kp:
            b2 = torch.zeros(bt.shape[0], Kp, dtype=bt.dtype, device=bt.device)
            b2[:, :bt.shape[1]] = bt
            bt = b2
        if a.shape[1] != Kp:
            a2 = torch.zeros(a.shape[0], Kp, dtype=a.dtype, device=a.device)
            a2[:, :a.shape[1]] = a
            a = a2
    res = gemm(a, bt, out=out, out_dtype=out_dtype, alpha=alpha, act=act, aux=aux, m_dev=m_dev, accumulate=accumulate)
    return res if out is not None else res[:, :N]


def transpose(x, out_dtype=None, row_scale=None, want_plain=False, want_t=True):
    """-> (x^T [C, R padded to a chunk multiple, zero filled], plain copy or None)."""
    R, C = x.shape
    od = out_dtype or x.dtype
    v = _vec(od)
    xt = None
    if want_t:
        Rp = _round_up(R, v)
        xt = (torch.zeros if Rp != R else torch.empty)(C, Rp, dtype=od, device=x.device)
    plain = torch.empty(R, C, dtype=od, device=x.device) if want_plain else None
    call('tell_transpose', x, x.stride(0), hip.dt(x), xt, xt.stride(0) if want_t else 0, plain,
         plain.stride(0) if want_plain else 0, hip.dt(od), row_scale, R, C)
    return xt, plain


def zeros_group(specs, device):
    """[(shape, dtype), ...] -> zero tensors carved from ONE allocation cleared by ONE launch (a training step clears a
    dozen scratch buffers - count-limited gathers / GEMMs only write their first `count` rows)."""
    sizes = []
    for shape, dtype in specs:
        n = 1
        for d in shape:
            n *= d
        sizes.append(_round_up(n * torch.empty(0, dtype=dtype).element_size(), 256))
    flat = torch.empty(sum(sizes) // 4, dtype=torch.float32, device=device)
    if flat.is_cuda:
        call('tell_fill_f32', flat, flat.numel(), 0.0)
    else:
        flat.zero_()
    out, off = [], 0
    for (shape, dtype), nb in zip(specs, sizes):
        n = 1
        for d in shape:
            n *= d
        out.append(flat[off // 4:(off + nb) // 4].view(dtype)[:n].view(*shape))
        off += nb
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    y = torch.empty_like(x, dtype=dtype)
    call('tell_cast', x, hip.dt(x), y, hip.dt(y), x.numel())
    return y


# Deferred column sums: gradients that only the optimizer reads (LayerNorm gamma / beta from per-row-block partials,
# bias_k / bias_v of the attentions from per-batch rows) are folded by ONE launch when the backward pass is over
# (tell_colsum_multi) instead of one or two ~5 us launches each on the critical stream: the trainer turns the queue on
# around backward, like the weight-gradient queue.
_FINISH = {'defer': False, 'jobs': []}


def finish_defer(on):
    _FINISH['defer'] = bool(on)


def finish_drop():
    _FINISH['jobs'] = []


def finish_job(src, dst0, w0, dst1):
    """dst0[c] += sum_r src[r][c] (c < w0), dst1[c - w0] += ... (c >= w0).  src: fp32 [rows, width]; queued while
    deferring (the tensors are kept alive by the queue), else done now."""
    _FINISH['jobs'].append((src, dst0, int(w0), dst1))
    if not _FINISH['defer']:
        finish_flush()


def finish_flush():
    jobs, _FINISH['jobs'] = _FINISH['jobs'], []
    if not jobs:
        return
    # the kernel accumulates with a plain read-modify-write (one block per destination column range): two jobs of ONE
    # launch that share a destination (a tied LayerNorm, gradient accumulation over several passes) would race, so a
    # job whose destination is already in this launch goes into the next one (launches are ordered on the stream)
    seen, later, now = set(), [], []
    for j in jobs:
        keys = {t.data_ptr() for t in (j[1], j[3]) if t is not None}
        if keys & seen:
            later.append(j)
        else:
            seen |= keys
            now.append(j)
    if later:
        _FINISH['jobs'] = later
    jobs = now
    n = len(jobs)
    call('tell_colsum_multi', n, _ptr_array([j[0] for j in jobs]), (ctypes.c_long * n)(*[j[0].stride(0) for j in jobs]),
         _int_array([j[0].shape[0] for j in jobs]), _int_array([j[0].shape[1] for j in jobs]),
         _int_array([j[2] for j in jobs]),
         (ctypes.c_void_p * n)(*[j[1].data_ptr() if j[1] is not None else None for j in jobs]),
         (ctypes.c_void_p * n)(*[j[3].data_ptr() if j[3] is not None else None for j in jobs]))
    if later:
        finish_flush()


def colsum_into(x2, out, scale=1.0, m_dev=None):
    ws = torch.empty(hip.lib().tell_colsum_chunks(x2.shape[0]) * x2.shape[1], dtype=torch.float32, device=x2.device)
    call('tell_colsum', x2, x2.stride(0), x2.shape[0], x2.shape[1], hip.dt(x2), out, 1, m_dev, float(scale), ws)


# --------------------------------------------------------------------------- #
# working copies of weights (compute dtype), cached per weights epoch
# --------------------------------------------------------------------------- #
_wcache = {}


def _stamp(p):
    # trainable masters are rewritten by the optimizer kernel (which bypasses torch's version counter) ->
    # keyed on the weights epoch; frozen tensors (encoders, buffers) only change through torch ops
    epoch = rt.weights_epoch() if p.requires_grad else -1
    if p.requires_grad:
        rt.wait_weight_update()          # an optimizer step may still be in flight on the update stream
    return (epoch, rt.compute_dtype(), p._version, p.data_ptr())


def _fresh(p, key):
    """The cached working copy of p under `key`, or None when there is none / it is stale."""
    e = _wcache.get((id(p), key))
    # id() is only unique among LIVE objects: the entry remembers its tensor weakly, so a new tensor that
    # inherited a dead one's id (and possibly its storage address) never sees the old working copy
    if e is None or e[0] != _stamp(p) or e[2]() is not p:
        return None
    return e


def _cached(p, key, maker):
    e = _fresh(p, key)
    if e is None:
        e = (_stamp(p), maker(), weakref.ref(p))
        _wcache[(id(p), key)] = e
    return e[1]


def clear_weight_cache():
    _wcache.clear()


def drop_trainable_cache():
    """Forget the working copies derived from TRAINABLE parameters (weight-normalised weights, the concatenated
    softmax head).  The copies of frozen tensors stay: captured encoder graphs hold their addresses."""
    for k in [k for k, e in _wcache.items() if e[2]() is None or e[2]().requires_grad]:
        del _wcache[k]


def weight(p, rows=None):
    """Working copy [N,K] (compute dtype) of fp32 parameter rows p[r0:r1]."""
    sh = getattr(p, '_tell_shadow', None)
    if sh is not None and rt.compute_dtype() == torch.bfloat16:
        # the optimizer kernel maintains a bf16 copy of the flat parameters (training/optimizers.py); anything else
        # that wrote the parameter (load_state_dict, broadcast) bumped its version -> refresh that slice once
        rt.wait_weight_update()
        if p._tell_shadow_version != p._version:
            call('tell_cast', p.detach().contiguous(), hip.F32, sh, hip.BF16, p.numel())
            p._tell_shadow_version = p._version
        src = sh if rows is None else sh[rows[0]:rows[1]]
        return src.reshape(src.shape[0], -1)

    def make():
        src = p.detach() if rows is None else p.detach()[rows[0]:rows[1]]
        src = src.reshape(src.shape[0], -1)
        return cast(src, rt.compute_dtype()) if rt.compute_dtype() != torch.float32 else src.contiguous()
    return _cached(p, ('w', rows), make)


def weight_t(p, rows=None):
    """Transposed working copy [K, N padded] for dX = dY . W."""
    def make():
        src = p.detach() if rows is None else p.detach()[rows[0]:rows[1]]
        return transpose(src.reshape(src.shape[0], -1), out_dtype=rt.compute_dtype())[0]
    return _cached(p, ('wt', rows), make)


def wn_weight(g, v):
    """Weight-normalised working weight (tell/modules/linear.py:33): (w [N,K] compute dtype, norms)."""
    def make():
        R, C = v.shape
        norms = torch.empty(R, dtype=torch.float32, device=v.device)
        w = torch.empty(R, C, dtype=rt.compute_dtype(), device=v.device)
        call('tell_wn_weight', g.detach(), v.detach(), R, C, w, hip.dt(w), norms)
        return w, norms
    return _cached(v, ('wn', g._version, g.data_ptr()), make)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(values):
    return (ctypes.c_int * len(values))(*values)


def wn_prepare(pairs):
    """Working weights of MANY GehringLinears [(weight_g, weight_v), ...] in one launch: whatever is stale in the cache
    is rebuilt together (tell_wn_weight_multi), so the layers' own wn_weight() calls all hit.  The decoder calls this
    once per forward: 20 launches of ~10 us become one."""
    stale = []
    for g, v in pairs:
        if v.is_cuda and _fresh(v, ('wn', g._version, g.data_ptr())) is None:
            stale.append((g, v))
    if len(stale) < 2:
        return
    dtype = rt.compute_dtype()
    ws = [torch.empty(v.shape[0], v.shape[1], dtype=dtype, device=v.device) for _, v in stale]
    norms = [torch.empty(v.shape[0], dtype=torch.float32, device=v.device) for _, v in stale]
    call('tell_wn_weight_multi', len(stale), _ptr_array([g.detach() for g, _ in stale]),
         _ptr_array([v.detach() for _, v in stale]), _ptr_array(ws), _ptr_array(norms),
         _int_array([v.shape[0] for _, v in stale]), _int_array([v.shape[1] for _, v in stale]), hip.dt(dtype))
    for (g, v), w, n in zip(stale, ws, norms):
        _wcache[(id(v), ('wn', g._version, g.data_ptr()))] = (_stamp(v), (w, n), weakref.ref(v))


_WN_PENDING = {'defer': False, 'items': []}


def wn_defer(on):
    """While on, the weight-norm chain rule of every GehringLinear (dW -> dg, dv) is queued instead of launched; the
    caller flushes the queue with wn_flush() once the backward pass is over (trainer._backward)."""
    _WN_PENDING['defer'] = bool(on)


def wn_drop():
    _WN_PENDING['items'] = []


def wn_flush():
    items, _WN_PENDING['items'] = _WN_PENDING['items'], []
    if items:
        _wn_backward(items)


def _wn_backward(items):
    """dW -> dg, dv of [(dW, g, v, norms)] in one launch."""
    for _, g, v, _ in items:
        note_grad_write(g)
        note_grad_write(v)
    # dg / dv of a GehringLinear used once per step are stored, not accumulated (see _GRAD_STORE)
    store = [int(grad_stored(g) and grad_stored(v)) for _, g, v, _ in items]
    call('tell_wn_backward_multi2', len(items), _ptr_array([i[0] for i in items]),
         _ptr_array([i[1].detach() for i in items]), _ptr_array([i[2].detach() for i in items]),
         _ptr_array([i[3] for i in items]), _int_array([i[2].shape[0] for i in items]),
         _int_array([i[2].shape[1] for i in items]), _ptr_array([grad_buffer(i[1], True) for i in items]),
         _ptr_array([grad_buffer(i[2], True) for i in items]), _int_array(store))


def wn_weight_t(g, v):
    """w^T of wn_weight (only the fp32 / unaligned fallback of the dgrad GEMM needs it)."""
    return _cached(v, ('wn_t', g._version, g.data_ptr()), lambda: transpose(wn_weight(g, v)[0])[0])


# --------------------------------------------------------------------------- #
# weight gradients on a second stream
# --------------------------------------------------------------------------- #
# The backward chain of the decoder (M = 512 rows) is a string of small, latency-bound kernels; the weight-
# gradient GEMMs hang off it as leaves (only the optimizer reads their output).  They are issued on a side
# stream, fill the idle CUs underneath the dgrad chain, and are joined back when the backward pass ends.
# All writers of a Linear / GehringLinear gradient buffer go through here, so accumulations into a shared
# buffer (row slices of in_proj_weight, several uses of one layer) stay ordered on that one stream.
_WGRAD = {'enabled': os.environ.get('TELL_WGRAD_STREAM', '0') == '1', 'streams': {}, 'main': None, 'queue': [],
          'hooked': False}
_WGRAD_FLUSH = int(os.environ.get('TELL_WGRAD_FLUSH', '6'))            # deferred jobs per hand-over (one event + one stream switch per batch, not per GEMM)


def _flush_wgrad():
    q = _WGRAD['queue']
    if not q:
        return
    main = _WGRAD['main']
    dev = main.device
    side = _WGRAD['streams'].get(dev)
    if side is None:
        side = _WGRAD['streams'][dev] = streams.get('wgrad', dev)
    side.wait_stream(main)                          # every queued input was produced on `main` before now
    with torch.cuda.stream(side), hip.bound_stream():
        for job, keep in q:
            job()
            for t in keep:                          # inputs were allocated on `main`: keep them for `side`
                if t is not None:
                    t.record_stream(side)
    _WGRAD['queue'] = []
    _WGRAD['dirty'] = True


def _join_wgrad():
    _WGRAD['hooked'] = False
    _flush_wgrad()
    if _WGRAD.get('dirty'):
        _WGRAD['dirty'] = False
        main = _WGRAD['main']
        main.wait_stream(_WGRAD['streams'][main.device])


def wgrad_job(job, *inputs):
    """Queue `job` (launches that write weight/bias gradient buffers from `inputs`) for the weight-gradient
    stream; without a GPU side stream it runs in place."""
    if not _WGRAD['enabled'] or not inputs[0].is_cuda:
        job()
        return
    main = torch.cuda.current_stream(inputs[0].device)
    if _WGRAD['main'] is not None and _WGRAD['main'] != main and (_WGRAD['queue'] or _WGRAD.get('dirty')):
        _join_wgrad()                               # the caller switched streams: finish the old hand-over first
    _WGRAD['main'] = main
    _WGRAD['queue'].append((job, inputs))
    if not _WGRAD['hooked']:
        try:                                        # inside backward(): join when the engine finishes the pass
            torch.autograd.Variable._execution_engine.queue_callback(_join_wgrad)
            _WGRAD['hooked'] = True
        except RuntimeError:                        # called outside a backward pass: run now, the caller joins
            _flush_wgrad()
            return
    if len(_WGRAD['queue']) >= _WGRAD_FLUSH:
        _flush_wgrad()


class _GradReadyFn(Function):
    """Identity whose backward reports that everything downstream of it has finished its backward."""

    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        cb = rt.grad_ready_callback()
        if cb is not None:
            cb(ctx.tag)
        return g, None


def grad_ready_marker(x, tag):
    """Put in front of a block of layers: when its backward runs, the gradients of that block's parameters are final
    (data-parallel bucketed all-reduce, training/trainer.py).  A no-op unless a callback is registered."""
    if rt.grad_ready_callback() is None or not (torch.is_grad_enabled() and x.requires_grad):
        return x
    return _GradReadyFn.apply(x, tag)


def flush_wgrad_stream():
    """Hand every queued weight-gradient job to the side stream now; -> that stream (None if unused)."""
    _flush_wgrad()
    main = _WGRAD['main']
    return None if main is None else _WGRAD['streams'].get(main.device)


def join_wgrad_stream():
    """Make the producing stream wait for outstanding weight-gradient work (no-op if there is none)."""
    if _WGRAD['main'] is not None:
        _join_wgrad()


# --------------------------------------------------------------------------- #
# Linear layers
# --------------------------------------------------------------------------- #
class LinearFn(Function):
    """y = act((x W[rows]^T + b[rows]) * alpha); plain (xavier) weights: attention
    projections (multi_head.py:488-526), DynamicConv tap projection (dynamic.py:300)."""

    @staticmethod
    def forward(ctx, x, w_param, b_param, rows, act, alpha, need_dx, x_t=None, b_rows=None):
        x2 = as2d(x)
        w = weight(w_param, rows)
        if b_rows is None:
            b_rows = rows
        b = None
        if b_param is not None:
            b = b_param.detach() if b_rows is None else b_param.detach()[b_rows[0]:b_rows[1]]
        y = gemm(x2, w, bias=b, bias_mode=1 if b is not None else 0, act=act, alpha=alpha)
        ctx.save_for_backward(x2, y if act == 1 else None, x_t)
        ctx.meta = (w_param, b_param, rows, act, alpha, need_dx, x.shape, b_rows)
        return y.view(*x.shape[:-1], y.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x2, y, x_t = ctx.saved_tensors
        w_param, b_param, rows, act, alpha, need_dx, xshape, b_rows = ctx.meta
        dy2 = as2dc(dy) if act == 1 else as2d(dy)
        if act == 1:
            d = torch.empty_like(dy2)
            call('tell_relu_bwd', dy2, y, d, dy2.numel(), hip.dt(dy2))
            dy2 = d
        r0, r1 = rows if rows is not None else (0, w_param.shape[0])
        gb = None
        if b_param is not None and b_param.requires_grad:
            gb = grad_buffer(b_param)
            gb = gb if b_rows is None else gb[b_rows[0]:b_rows[1]]
        if w_param.requires_grad or gb is not None:
            def job(gb=gb):
                if w_param.requires_grad:
                    gw2, acc = wgrad_target(w_param, rows)
                    if x_t is not None:              # fp32 parity mode: the caller shared one transpose of x
                        gemm(transpose(dy2)[0], x_t, out=gw2, alpha=alpha, accumulate=acc)
                    else:                            # the bias gradient rides on the wgrad GEMM's A tiles
                        gemm_tn(dy2, x2, out=gw2, alpha=alpha, accumulate=acc, asum=gb, asum_scale=alpha)
                        gb = None
                if gb is not None:
                    colsum_into(dy2, gb, scale=alpha)
            wgrad_job(job, dy2, x2, x_t)
        dx = None
        if need_dx:
            dx = gemm_nn(dy2, weight(w_param, rows), b_t=lambda: weight_t(w_param, rows), alpha=alpha)
            dx = dx.reshape(xshape) if dx.is_contiguous() else dx.contiguous().view(xshape)
        return dx, None, None, None, None, None, None, None, None


class GroupedLinearFn(Function):
    """n independent plain linears y_i = (x_i W_i[rows_i]^T + b_i[b_rows_i]) * alpha_i as ONE launch forward
    (tell_gemm_grouped, NT form) and one launch for the input gradients (B K-major form); the weight gradients go
    through gemm_tn - queued with everything else while the trainer defers them.  The decoder's context block uses it
    for the query projections and the output projections of its attentions (multi_head.py:488-526, 4 per layer each).
    bf16 path only; specs: [(w_param, rows, b_param, b_rows, alpha)]."""

    @staticmethod
    def forward(ctx, specs, *xs):
        x2s, ys, probs = [], [], []
        for x, (w_param, rows, b_param, b_rows, alpha) in zip(xs, specs):
            x2 = as2d(x)
            w = weight(w_param, rows)
            b = None
            if b_param is not None:
                b = b_param.detach() if b_rows is None else b_param.detach()[b_rows[0]:b_rows[1]]
            y = torch.empty(x2.shape[0], w.shape[0], dtype=x2.dtype, device=x2.device)
            probs.append(dict(a=x2, b=w, out=y, form='nt', bias=b, alpha=alpha))
            x2s.append(x2)
            ys.append(y.view(*x.shape[:-1], y.shape[1]))
        gemm_grouped(probs)
        ctx.save_for_backward(*x2s)
        ctx.specs = specs
        ctx.shapes = [x.shape for x in xs]
        ctx.need_dx = [x.requires_grad for x in xs]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x2s = ctx.saved_tensors
        dxs, probs = [], []
        for x2, dy, shape, need_dx, (w_param, rows, b_param, b_rows, alpha) in zip(x2s, dys, ctx.shapes, ctx.need_dx,
                                                                                  ctx.specs):
            if dy is None:
                dxs.append(None)
                continue
            dy2 = as2d(dy)
            if not _kmajor_ok(dy2):
                dy2 = dy2.contiguous()
            r0, r1 = rows if rows is not None else (0, w_param.shape[0])
            gb = None
            if b_param is not None and b_param.requires_grad:
                gb = grad_buffer(b_param)
                gb = gb if b_rows is None else gb[b_rows[0]:b_rows[1]]
            if w_param.requires_grad:
                gw2, acc = wgrad_target(w_param, rows)
                gemm_tn(dy2, x2, out=gw2, alpha=alpha, accumulate=acc, asum=gb, asum_scale=alpha)
            elif gb is not None:
                colsum_into(dy2, gb, scale=alpha)
            if need_dx:
                w = weight(w_param, rows)
                dx = torch.empty(dy2.shape[0], w.shape[1], dtype=dy2.dtype, device=dy2.device)
                probs.append(dict(a=dy2, b=w, out=dx, form='nn', alpha=alpha))
                dxs.append(dx.view(shape))
            else:
                dxs.append(None)
        gemm_grouped(probs)
        return (None,) + tuple(dxs)


def grouped_linear(xs, specs):
    """[y_i] for [x_i] and specs [(w_param, rows, b_param, b_rows, alpha)]: one launch (bf16 CUDA tensors with rows of
    whole 16-byte chunks), else one ops.linear per item."""
    ok = rt.compute_dtype() == torch.bfloat16 and all(
        x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0 and as2d(x).stride(1) == 1 and
        as2d(x).stride(0) % 8 == 0 and as2d(x).data_ptr() % 16 == 0 for x in xs)
    if not ok or len(xs) < 2:
        return [linear(x, w, b, rows=rows, alpha=alpha, b_rows=b_rows) for x, (w, rows, b, b_rows, alpha) in zip(xs, specs)]
    return list(GroupedLinearFn.apply(specs, *xs))


def linear(x, w_param, b_param=None, rows=None, act=0, alpha=1.0, x_t=None, b_rows=None):
    """x_t: optional precomputed transpose of as2d(x) (shared by several projections of one context)."""
    return LinearFn.apply(x, w_param, b_param, rows, act, alpha, x.requires_grad, x_t, b_rows)



def _adjacent(a, b):
    """b starts where a ends (same dtype, both dense): the two live back to back in one flat buffer."""
    return (a is not None and b is not None and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous() and
            a.data_ptr() + a.numel() * a.element_size() == b.data_ptr() and
            a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())


def _stacked(a, b):
    """[rows_a + rows_b, C] view over two adjacent [rows, C] tensors (no copy)."""
    C = a.shape[-1]
    return torch.as_strided(a, (a.numel() // C + b.numel() // C, C), (C, 1))


class KVLinearFn(Function):
    """K and V projections of one context (multi_head.py:500-518) as ONE GEMM with N = 2E:
    kv[rows, 0:E] = x Wk^T + b_k, kv[rows, E:2E] = x Wv^T + b_v.  The two weights (slices E:3E of in_proj_weight, or
    k_proj_weight / v_proj_weight, which the flat parameter buffer stores back to back) are used as one [2E, kdim]
    operand; so are their gradients.  bf16 path (the fp32 parity mode keeps the two plain projections)."""

    @staticmethod
    def forward(ctx, x, wk, rk, wv, rv, b_param, E, need_dx):
        x2 = as2d(x)
        a, b = weight(wk, rk), weight(wv, rv)
        w = _stacked(a, b) if _adjacent(a, b) else _cached(wk, ('kv', rk, id(wv), wv._version), lambda: torch.cat([a, b], 0))
        bias = b_param.detach()[E:3 * E]
        y = gemm(x2, w, bias=bias, bias_mode=1)
        ctx.save_for_backward(x2)
        ctx.meta = (wk, rk, wv, rv, b_param, E, need_dx, x.shape, w)
        return y.view(*x.shape[:-1], 2 * E)

    @staticmethod
    def backward(ctx, dy):
        x2, = ctx.saved_tensors
        wk, rk, wv, rv, b_param, E, need_dx, xshape, w = ctx.meta
        dy2 = as2d(dy)
        gb = grad_buffer(b_param)[E:3 * E] if b_param.requires_grad else None
        if wk.requires_grad:
            (gk, acc_k), (gv, acc_v) = wgrad_target(wk, rk), wgrad_target(wv, rv)

            def job(gb=gb):
                kv_wgrad(dy2, x2, gk, acc_k, gv, acc_v, gb, E)
            wgrad_job(job, dy2, x2)
        elif gb is not None:
            colsum_into(dy2, gb)
        dx = None
        if need_dx:
            # the article's dX (16384 x 1024 from K = 2048): as an NT product against the transposed stacked weight it
            # runs on the 256x256 ping-pong kernel (transpose 14 us + 60 us against 97 us for the K-major 128x128 form,
            # which is bound by operand re-reads through L2); gemm_nn only takes it for that size class
            dx = gemm_nn(dy2, w, b_t=lambda: _cached(wk, ('kv_t', rk, id(wv), wv._version), lambda: transpose(w)[0]))
            dx = dx.reshape(xshape) if dx.is_contiguous() else dx.contiguous().view(xshape)
        return dx, None, None, None, None, None, None, None


def kv_wgrad(dy2, x2, gk, acc_k, gv, acc_v, gb, E):
    """Weight gradients of a stacked K / V projection: one product when the two gradient row blocks are neighbours in
    the flat buffer (and agree on accumulate / store), else two."""
    if _adjacent(gk, gv) and acc_k == acc_v:
        gemm_tn(dy2, x2, out=_stacked(gk, gv), accumulate=acc_k, asum=gb)
    else:
        gemm_tn(dy2[:, :E], x2, out=gk, accumulate=acc_k, asum=None if gb is None else gb[:E])
        gemm_tn(dy2[:, E:], x2, out=gv, accumulate=acc_v, asum=None if gb is None else gb[E:])


def kv_linear(x, wk, rk, wv, rv, b_param, E):
    return KVLinearFn.apply(x, wk, rk, wv, rv, b_param, E, x.requires_grad)


class WNLinearFn(Function):
    """GehringLinear (tell/modules/linear.py:8-33): y = act(x (g v/||v||)^T + b)."""

    @staticmethod
    def forward(ctx, x, g, v, b, act, need_dx):
        x2 = as2d(x)
        w, norms = wn_weight(g, v)
        y = gemm(x2, w, bias=b.detach() if b is not None else None, bias_mode=1 if b is not None else 0,
                 act=act)
        ctx.save_for_backward(x2, y if act == 1 else None)
        ctx.meta = (g, v, b, act, need_dx, x.shape)
        return y.view(*x.shape[:-1], y.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x2, y = ctx.saved_tensors
        g, v, b, act, need_dx, xshape = ctx.meta
        dy2 = as2dc(dy) if act == 1 else as2d(dy)
        if act == 1:
            d = torch.empty_like(dy2)
            call('tell_relu_bwd', dy2, y, d, dy2.numel(), hip.dt(dy2))
            dy2 = d
        w, norms = wn_weight(g, v)
        gb = grad_buffer(b) if (b is not None and b.requires_grad) else None
        if v.requires_grad or gb is not None:
            def job(gb=gb):
                if v.requires_grad:
                    dW = gemm_tn(dy2, x2, out_dtype=torch.float32, asum=gb)
                    gb = None
                    if _WN_PENDING['defer']:
                        _WN_PENDING['items'].append((dW, g, v, norms))
                    else:
                        _wn_backward([(dW, g, v, norms)])
                if gb is not None:
                    colsum_into(dy2, gb)
            wgrad_job(job, dy2, x2)
        dx = None
        if need_dx:
            dx = gemm_nn(dy2, w, b_t=lambda: wn_weight_t(g, v))
            dx = dx.reshape(xshape) if dx.is_contiguous() else dx.contiguous().view(xshape)
        return dx, None, None, None, None, None


def wn_wgrad(dy2, x2, g, v, b, norms):
    """Weight / bias gradients of a GehringLinear from dY [rows, out] and X [rows, in] (both as the forward pass left
    them): dW into a fresh fp32 tile set (queued with the pass's other weight gradients), the weight-norm chain rule
    dW -> dg, dv deferred to the pass's single wn_backward launch, the bias gradient on the GEMM's A tiles."""
    gb = grad_buffer(b) if (b is not None and b.requires_grad) else None
    if v.requires_grad:
        dW = gemm_tn(dy2, x2, out_dtype=torch.float32, asum=gb)
        if _WN_PENDING['defer']:
            _WN_PENDING['items'].append((dW, g, v, norms))
        else:
            _wn_backward([(dW, g, v, norms)])
    elif gb is not None:
        colsum_into(dy2, gb)


def linear_wgrad(dy2, x2, w_param, rows=None, b_param=None, b_rows=None, alpha=1.0):
    """Weight / bias gradients of a plain linear (rows of w_param), accumulated into the flat gradient buffer."""
    r0, r1 = rows if rows is not None else (0, w_param.shape[0])
    gb = None
    if b_param is not None and b_param.requires_grad:
        gb = grad_buffer(b_param)
        gb = gb if b_rows is None else gb[b_rows[0]:b_rows[1]]
    if w_param.requires_grad:
        gw2, acc = wgrad_target(w_param, rows)
        gemm_tn(dy2, x2, out=gw2, alpha=alpha, accumulate=acc, asum=gb, asum_scale=alpha)
    elif gb is not None:
        colsum_into(dy2, gb, scale=alpha)


def wn_linear(x, g, v, b=None, act=0):
    return WNLinearFn.apply(x, g, v, b, act, x.requires_grad)


# --------------------------------------------------------------------------- #
# elementwise
# --------------------------------------------------------------------------- #
class FanOutFn(Function):
    """x -> n aliases of x; backward adds the n incoming gradients in one kernel (autograd would issue n-1 adds)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if len(gs) == 1:
            return gs[0], None
        gs = [g if g.is_contiguous() else g.contiguous() for g in gs]
        vec = _vec(gs[0].dtype)
        if len(gs) > 8 or gs[0].numel() % vec or any(g.data_ptr() % 16 for g in gs) or not gs[0].is_cuda:
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(gs[0])
        ptrs = gs + [None] * (8 - len(gs))
        call('tell_sum_n', *ptrs, len(gs), out, out.numel(), hip.dt(out))
        return out, None


def fan_out(x, n):
    """n handles on x for n consumers (see FanOutFn); plain aliases when no gradient is needed."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return FanOutFn.apply(x, n)


class GLUFn(Function):
    @staticmethod
    def forward(ctx, h):
        h2 = as2dc(h)
        C = h2.shape[1] // 2
        y = torch.empty(h2.shape[0], C, dtype=h.dtype, device=h.device)
        call('tell_glu_fwd', h2, y, h2.shape[0], C, hip.dt(h2))
        ctx.save_for_backward(h2)
        ctx.shape = h.shape
        return y.view(*h.shape[:-1], C)

    @staticmethod
    def backward(ctx, dy):
        h2, = ctx.saved_tensors
        dy2 = as2dc(dy)
        dh = torch.empty_like(h2)
        call('tell_glu_bwd', h2, dy2, dh, h2.shape[0], h2.shape[1] // 2, hip.dt(h2))
        return dh.view(ctx.shape)


def glu(h):
    return GLUFn.apply(h)


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, salt):
        x = x.contiguous()
        y = torch.empty_like(x)
        call('tell_dropout', x, y, x.numel(), float(p), rt.seed(), salt, hip.dt(x))
        ctx.p, ctx.salt = p, salt
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call('tell_dropout', dy, dx, dy.numel(), float(ctx.p), rt.seed(), ctx.salt, hip.dt(dy))
        return dx, None, None


def dropout(x, p, training, salt=None):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p, rt.next_salt() if salt is None else salt)


class LayerNormFn(Function):
    """y = LayerNorm(res + dropout(x)) (post-LN residual blocks,
    decoder_faces_objects.py:263-266 and :283-287)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p, salt):
        x2 = as2d(x)
        r2 = as2d(res) if res is not None else None
        rows, C = x2.shape
        y = torch.empty(rows, C, dtype=x.dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        call('tell_layernorm_fwd', x2, x2.stride(0), r2, r2.stride(0) if r2 is not None else 0,
             gamma.detach(), beta.detach(), y, y.stride(0), mean, rstd, rows, C, float(eps), float(p),
             rt.seed(), salt, hip.dt(x2))
        ctx.save_for_backward(x2, r2, mean, rstd)
        ctx.meta = (gamma, beta, p, salt, x.shape, res is not None and res.requires_grad, x.requires_grad)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, mean, rstd = ctx.saved_tensors
        gamma, beta, p, salt, shape, need_dres, need_dx = ctx.meta
        dy2 = as2d(dy)
        rows, C = x2.shape
        nb = hip.lib().tell_layernorm_bwd_blocks(rows)
        partial = torch.empty(nb * 2 * C, dtype=torch.float32, device=x2.device)
        dx = torch.empty_like(x2) if need_dx else None
        same = (p <= 0) and need_dx
        dres = None
        if need_dres and r2 is not None:
            dres = dx if same else torch.empty_like(x2)
        defer = _FINISH['defer'] and x2.is_cuda
        call('tell_layernorm_bwd', dy2, dy2.stride(0), x2, x2.stride(0), r2,
             r2.stride(0) if r2 is not None else 0, gamma.detach(), mean, rstd,
             dx, dx.stride(0) if dx is not None else 0,
             None if (dres is None or same) else dres, dres.stride(0) if dres is not None else 0, 0,
             None if defer else grad_buffer(gamma), None if defer else grad_buffer(beta), 1, partial, rows, C, float(p),
             rt.seed(), salt, hip.dt(x2))
        if defer:
            finish_job(partial.view(nb, 2 * C), grad_buffer(gamma), C, grad_buffer(beta))
        return (dx.view(shape) if dx is not None else None,
                dres.view(shape) if dres is not None else None, None, None, None, None, None)


def layer_norm(x, res, gamma, beta, eps=1e-5, p=0.0, training=False):
    p = p if training else 0.0
    return LayerNormFn.apply(x, res, gamma, beta, eps, p, rt.next_salt() if p > 0 else 0)


_LN_CAT = os.environ.get('TELL_LN_CAT', '1') != '0'          # A/B aid: 0 = one launch per LayerNorm


class LNCatFn(Function):
    """cat_i LayerNorm_i(res + dropout(x_i)) along the feature axis: the n context branches of a decoder layer
    (decoder_faces_objects.py:283-352) write their normalised outputs straight into the [rows, n*C] input of
    context_fc (:354, no torch.cat), and in backward the n residual gradients are accumulated into ONE buffer by the
    LayerNorm kernels themselves (no pairwise adds)."""

    @staticmethod
    def forward(ctx, res, eps, p, salts, n, *args):
        xs, gammas, betas = args[:n], args[n:2 * n], args[2 * n:3 * n]
        r2 = as2d(res)
        rows, C = r2.shape
        cat = torch.empty(rows, n * C, dtype=res.dtype, device=res.device)
        mean = torch.empty(n, rows, dtype=torch.float32, device=res.device)
        rstd = torch.empty(n, rows, dtype=torch.float32, device=res.device)
        x2s = [as2d(x) for x in xs]
        one = LNCatFn._one_launch(r2, x2s, C, n)
        if one:
            call('tell_layernorm_cat_fwd', n, _ptr_array(x2s), x2s[0].stride(0), r2, r2.stride(0),
                 _ptr_array([g.detach() for g in gammas]), _ptr_array([b.detach() for b in betas]), cat, cat.stride(0),
                 mean, rstd, rows, C, float(eps), float(p), rt.seed(), (ctypes.c_uint32 * n)(*salts), hip.dt(r2))
        for i in range(0 if one else n):
            y = cat[:, i * C:(i + 1) * C]
            call('tell_layernorm_fwd', x2s[i], x2s[i].stride(0), r2, r2.stride(0), gammas[i].detach(), betas[i].detach(),
                 y, y.stride(0), mean[i], rstd[i], rows, C, float(eps), float(p), rt.seed(), salts[i], hip.dt(r2))
        ctx.save_for_backward(r2, mean, rstd, *x2s)
        ctx.meta = (gammas, betas, p, salts, n, res.shape, res.requires_grad, [x.requires_grad for x in xs])
        return cat.view(*res.shape[:-1], n * C)

    @staticmethod
    def _one_launch(r2, x2s, C, n):
        """All n LayerNorms in one launch (tell_layernorm_cat_*): bf16, C = 512 / 1024, one row stride, 16-byte rows."""
        return (_LN_CAT and r2.is_cuda and r2.dtype == torch.bfloat16 and C in (512, 1024) and 2 <= n <= 8 and
                r2.stride(0) % 8 == 0 and r2.data_ptr() % 16 == 0 and
                all(x.stride(0) == x2s[0].stride(0) and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 for x in x2s))

    @staticmethod
    def backward(ctx, dcat):
        r2, mean, rstd = ctx.saved_tensors[:3]
        x2s = ctx.saved_tensors[3:]
        gammas, betas, p, salts, n, shape, need_dres, need_dx = ctx.meta
        rows, C = r2.shape
        d2 = as2d(dcat)
        nb = hip.lib().tell_layernorm_bwd_blocks(rows)
        dres = torch.empty_like(r2) if need_dres else None
        dxs = []
        if LNCatFn._one_launch(r2, x2s, C, n) and d2.stride(0) % 8 == 0 and d2.data_ptr() % 16 == 0:
            partial = torch.empty(n * nb * 2 * C, dtype=torch.float32, device=r2.device)
            dxl = [torch.empty_like(x2s[i]) if need_dx[i] else None for i in range(n)]
            have = [d for d in dxl if d is not None]
            defer = _FINISH['defer']
            call('tell_layernorm_cat_bwd', n, d2, d2.stride(0), _ptr_array(list(x2s)), x2s[0].stride(0), r2, r2.stride(0),
                 _ptr_array([g.detach() for g in gammas]), mean, rstd,
                 (ctypes.c_void_p * n)(*[d.data_ptr() if d is not None else None for d in dxl]),
                 have[0].stride(0) if have else 0, dres, dres.stride(0) if dres is not None else 0,
                 None if defer else _ptr_array([grad_buffer(g) for g in gammas]),
                 None if defer else _ptr_array([grad_buffer(b) for b in betas]), partial,
                 rows, C, float(p), rt.seed(), (ctypes.c_uint32 * n)(*salts), hip.dt(r2))
            if defer:
                pv = partial.view(n, nb, 2 * C)
                for i in range(n):
                    finish_job(pv[i], grad_buffer(gammas[i]), C, grad_buffer(betas[i]))
            dxs = [d.view(shape) if d is not None else None for d in dxl]
            return (dres.view(shape) if dres is not None else None, None, None, None, None, *dxs) + (None,) * (2 * n)
        for i in range(n):
            dy = d2[:, i * C:(i + 1) * C]
            partial = torch.empty(nb * 2 * C, dtype=torch.float32, device=r2.device)
            dx = torch.empty_like(x2s[i]) if need_dx[i] else None
            call('tell_layernorm_bwd', dy, dy.stride(0), x2s[i], x2s[i].stride(0), r2, r2.stride(0),
                 gammas[i].detach(), mean[i], rstd[i], dx, dx.stride(0) if dx is not None else 0,
                 dres, dres.stride(0) if dres is not None else 0, int(i > 0),
                 grad_buffer(gammas[i]), grad_buffer(betas[i]), 1, partial, rows, C, float(p), rt.seed(), salts[i],
                 hip.dt(r2))
            dxs.append(dx.view(shape) if dx is not None else None)
        return (dres.view(shape) if dres is not None else None, None, None, None, None, *dxs) + (None,) * (2 * n)


def layer_norm_cat(xs, res, lns, p=0.0, training=False):
    """[LN_i(res + dropout(x_i))]_i concatenated on the last axis; lns: the nn.LayerNorm modules."""
    p = p if training else 0.0
    n = len(xs)
    salts = tuple(rt.next_salt() if p > 0 else 0 for _ in range(n))
    return LNCatFn.apply(res, lns[0].eps, p, salts, n, *xs, *[ln.weight for ln in lns], *[ln.bias for ln in lns])


# --------------------------------------------------------------------------- #
# DynamicConv core
# --------------------------------------------------------------------------- #
class DynConvFn(Function):
    @staticmethod
    def forward(ctx, x, logits, H, K, p, salt):
        x = x.contiguous()
        logits = logits.contiguous()
        T, B, C = x.shape
        y = torch.empty_like(x)
        taps = torch.empty(T * B * H, K, dtype=torch.float32, device=x.device)
        call('tell_dynconv_fwd', x, logits, y, taps, T, B, H, K, C // H, float(p), rt.seed(), salt, hip.dt(x))
        ctx.save_for_backward(x, taps)
        ctx.meta = (H, K, p, salt)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, taps = ctx.saved_tensors
        H, K, p, salt = ctx.meta
        T, B, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dlogits = torch.empty(T, B, H * K, dtype=x.dtype, device=x.device)
        call('tell_dynconv_bwd', x, dy, taps, dx, 0, dlogits, T, B, H, K, C // H, float(p), rt.seed(), salt,
             hip.dt(x))
        return dx, dlogits, None, None, None, None


# --------------------------------------------------------------------------- #
# LSTM decoder pieces (tell/models/decoder_flattened_lstm.py, the GloVe/LSTM baseline)
# --------------------------------------------------------------------------- #
class LSTMCellFn(Function):
    """Gate non-linearities + state update of nn.LSTMCell from the two gate pre-activations ([B,4H] each)."""

    @staticmethod
    def forward(ctx, g1, g2, c_prev):
        g1, g2, c_prev = g1.contiguous(), g2.contiguous(), c_prev.contiguous()
        B, H = c_prev.shape
        h, c = torch.empty_like(c_prev), torch.empty_like(c_prev)
        gates = torch.empty(B, 4 * H, dtype=torch.float32, device=g1.device)
        call('tell_lstm_cell_fwd', g1, g2, c_prev, h, c, gates, B, H, hip.dt(g1))
        ctx.save_for_backward(gates, c, c_prev)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c, c_prev = ctx.saved_tensors
        B, H = c.shape
        dg = torch.empty(B, 4 * H, dtype=c.dtype, device=c.device)
        dcp = torch.empty_like(c)
        call('tell_lstm_cell_bwd', dh.contiguous() if dh is not None else None,
             dc.contiguous() if dc is not None else None, gates, c, c_prev, dg, dcp, B, H, hip.dt(c))
        return dg, dg, dcp


def lstm_cell(g1, g2, c_prev):
    return LSTMCellFn.apply(g1, g2, c_prev)


class DotAttnFn(Function):
    """AttentionLayer core (decoder_flattened_lstm.py:46-60): dot-product scores of one query per batch element against
    source_hids [L,B,D], key-padding mask [B,L], softmax over L, weighted sum.  -> (ctx [B,D], probs [L,B] fp32)."""

    @staticmethod
    def forward(ctx, x, src, mask):
        if src.stride(2) != 1:
            src = src.contiguous()
        x = x.contiguous()
        L, B, D = src.shape
        out = torch.empty(B, D, dtype=x.dtype, device=x.device)
        probs = torch.empty(L, B, dtype=torch.float32, device=x.device)
        call('tell_dot_attn_fwd', src, src.stride(0), src.stride(1), x, mask, out, probs, L, B, D, hip.dt(x))
        ctx.save_for_backward(src, probs, x)
        ctx.need_dsrc = src.requires_grad
        ctx.mark_non_differentiable(probs)
        ctx.set_materialize_grads(False)          # (no zero tensor for the unused probs gradient)
        return out, probs

    @staticmethod
    def backward(ctx, dctx, _dprobs):
        src, probs, x = ctx.saved_tensors
        L, B, D = src.shape
        if dctx is None:
            return None, None, None
        dx = torch.empty(B, D, dtype=src.dtype, device=src.device)
        dsrc = torch.empty(L, B, D, dtype=src.dtype, device=src.device) if ctx.need_dsrc else None
        call('tell_dot_attn_bwd', src, src.stride(0), src.stride(1), probs, dctx.contiguous(), x, dx, dsrc, L, B, D,
             hip.dt(src))
        return dx, dsrc, None


def dot_attention(x, src, mask):
    return DotAttnFn.apply(x, src, mask)


class TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        call('tell_tanh_fwd', x, y, x.numel(), hip.dt(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dx = torch.empty_like(y)
        call('tell_tanh_bwd', dy.contiguous(), y, dx, y.numel(), hip.dt(y))
        return dx


def tanh(x):
    return TanhFn.apply(x)


class StaticTapsFn(Function):
    """LightweightConv1dTBC (lightweight.py:186-188): the learned [H,1,K] tap parameter seen as per-position tap
    logits [T,B,H*K], so that the DynamicConv kernels serve `decoder_conv_type: lightweight` unchanged; backward sums
    the logit gradients over (t, b) straight into the parameter's gradient buffer."""

    @staticmethod
    def forward(ctx, w_param, T, B):
        w = weight(w_param).reshape(1, 1, -1)                     # [1,1,H*K] compute dtype
        ctx.w_param = w_param
        return w.expand(T, B, w.shape[2]).contiguous()

    @staticmethod
    def backward(ctx, dl):
        p = ctx.w_param
        if p.requires_grad:
            colsum_into(as2dc(dl), grad_buffer(p).view(-1))
        return None, None, None


def static_taps(w_param, T, B):
    return StaticTapsFn.apply(w_param, T, B)


def dynamic_conv(x, logits, H, K, p=0.0, training=False):
    p = p if training else 0.0
    return DynConvFn.apply(x, logits, H, K, p, rt.next_salt() if p > 0 else 0)


# --------------------------------------------------------------------------- #
# attention core
# --------------------------------------------------------------------------- #
def _kv_strides(t, S, B):
    """t: [S, B, E] view (any strides, last dim contiguous) -> (ptr tensor, s-stride, b-stride)."""
    assert t.stride(2) == 1
    return t.stride(0), t.stride(1)


def _bias_row(p, dtype):
    """bias_k / bias_v [1,1,E] as a flat row in `dtype`: the cached / optimizer-maintained working copy when the
    tensor runs in the global compute dtype, an explicit cast otherwise (unit tests drive the kernels directly)."""
    if p is None:
        return None
    if dtype == rt.compute_dtype():
        return weight(p).reshape(-1)
    return cast(p.detach().reshape(-1), dtype)


_DKV_TARGET = {}      # data_ptr of a packed K|V projection -> the gradient view its attention backward must fill


class AttnFn(Function):
    """softmax(q k^T + mask) v with the virtual bias_k/bias_v row and zero row.
    q: [T,B,E] (already scaled); k, v: [S,B,E] views; mask: [B,S] uint8 or None.
    v is None: `k` is the packed projection [S,B,2E] of kv_linear (K in columns 0:E, V in E:2E); the gradient comes
    back as one [S,B,2E] tensor too (no slice / concatenate nodes in the autograd graph)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, bias_k, bias_v, H, has_zero, p, salt):
        T, B, E = q.shape
        S = k.shape[0]
        D = E // H
        packed = v is None
        if packed:
            kv = k
            assert kv.stride(2) == 1 and kv.shape[2] == 2 * E
            ctx.kv_layout = (tuple(kv.shape), tuple(kv.stride()))
            ctx.kv_ptr = kv.data_ptr()
            k, v = kv[..., :E], kv[..., E:]
        if q.stride(2) != 1:
            q = q.contiguous()
        out = torch.empty(T, B, E, dtype=q.dtype, device=q.device)
        lse = torch.empty(B * H, T, dtype=torch.float32, device=q.device)
        bk = _bias_row(bias_k, q.dtype)
        bv = _bias_row(bias_v, q.dtype)
        if S == 0:      # empty context (multi_head.py:349-374): only the virtual rows remain
            k = v = q.new_zeros(1, B, E)
        call('tell_attn_fwd', q, k, v, out, lse, mask, bk, bv, B, H, T, S, D, q.stride(0), q.stride(1),
             k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
             int(has_zero), float(p), rt.seed(), salt, hip.dt(q))
        ctx.save_for_backward(q, k, v, out, lse, mask, bk, bv)
        ctx.meta = (bias_k, bias_v, H, has_zero, p, salt, S, packed)
        ctx.mark_non_differentiable(lse)
        ctx.set_materialize_grads(False)          # autograd would fill a zero [B*H,T] gradient for lse per call
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        q, k, v, out, lse, mask, bk, bv = ctx.saved_tensors
        bias_k, bias_v, H, has_zero, p, salt, S, packed = ctx.meta
        T, B, E = q.shape
        D = E // H
        if dout is None:
            return (None,) * 10
        dout = dout.contiguous()
        dq = torch.empty_like(q)
        # (every element is ASSIGNED by the workgroup of its (b, h) while it handles q block 0: no zero fill)
        dbk = dbv = dbkv = None
        if bk is not None:
            gbk, gbv = (grad_buffer(t).view(-1) if t.requires_grad else None for t in (bias_k, bias_v))
            if gbk is not None and gbv is not None and _adjacent(gbk, gbv):
                dbkv = torch.empty(B, 2 * E, dtype=torch.float32, device=q.device)     # one column sum for both
                dbk, dbv = dbkv[:, :E], dbkv[:, E:]
            else:
                dbk = torch.empty(B, E, dtype=torch.float32, device=q.device)
                dbv = torch.empty(B, E, dtype=torch.float32, device=q.device)
        # dk/dv share k's / v's strides (the kernel uses one stride set for K and dK);
        # empty_like keeps the strides of dense permuted views ([B,S,E] storage seen as [S,B,E])
        kc, vc = k, v
        dkv = None
        if packed and S > 0:
            # one [S,B,2E] gradient buffer laid out like the packed projection; the K / V halves are views of it.
            # A projection that computed several layers' K|V side by side (blocks.KVAllFn) has reserved this layer's
            # slice of ONE gradient buffer: the input gradient of all layers is then a single long-K GEMM
            kshape, kstride = ctx.kv_layout
            dkv = _DKV_TARGET.pop(ctx.kv_ptr, None)
            if dkv is None:
                dkv = torch.empty_strided(kshape, kstride, dtype=k.dtype, device=k.device)
        if dkv is not None:
            dk, dv = dkv[..., :E], dkv[..., E:]
        else:
            dk, dv = torch.empty_like(kc), torch.empty_like(vc)
            if dk.stride() != kc.stride():
                kc = k.contiguous()
                dk = torch.empty_like(kc)
            if dv.stride() != vc.stride():
                vc = v.contiguous()
                dv = torch.empty_like(vc)
        call('tell_attn_bwd', q, kc, vc, out, dout, lse, mask, bk, bv, dq, dk, dv, dbk, dbv, B, H, T, S, D,
             q.stride(0), q.stride(1), kc.stride(0), kc.stride(1), vc.stride(0), vc.stride(1),
             out.stride(0), out.stride(1), int(has_zero), float(p), rt.seed(), salt, hip.dt(q))
        if dbkv is not None and _FINISH['defer']:
            finish_job(dbkv, gbk, E, gbv)
        elif dbkv is not None:
            colsum_into(dbkv, torch.as_strided(gbk, (2 * E,), (1,)))
        elif bias_k is not None and bias_k.requires_grad:
            colsum_into(dbk, grad_buffer(bias_k).view(-1))
            colsum_into(dbv, grad_buffer(bias_v).view(-1))
        if packed:
            if S == 0:
                gk = None
            elif dkv is not None:
                gk = dkv
            else:
                gk = torch.cat([dk, dv], dim=-1)
            return dq, gk, None, None, None, None, None, None, None, None
        return dq, dk if S > 0 else None, dv if S > 0 else None, None, None, None, None, None, None, None


def attention_avg_weights(q, k, mask, bias_k, lse, H, has_zero=True):
    """Head-averaged attention weights [B,T,S'] (multi_head.py:478-482), recomputed from q, k and
    the saved log-sum-exp; evaluation / demo only (need_weights)."""
    T, B, E = q.shape
    S = k.shape[0]
    S_total = S + (1 if bias_k is not None else 0) + int(has_zero)
    w = torch.empty(B, T, S_total, dtype=torch.float32, device=q.device)
    bk = _bias_row(bias_k, q.dtype)
    call('tell_attn_avg_weights', q, k, lse, mask, bk, w, B, H, T, S, E // H, q.stride(0), q.stride(1),
         k.stride(0), k.stride(1), int(has_zero), hip.dt(q))
    return w


def attention(q, k, v, mask, bias_k, bias_v, H, has_zero=True, p=0.0, training=False, return_lse=False):
    p = p if training else 0.0
    out, lse = AttnFn.apply(q, k, v, mask, bias_k, bias_v, H, has_zero, p, rt.next_salt() if p > 0 else 0)
    return (out, lse) if return_lse else out


# --------------------------------------------------------------------------- #
# RoBERTa layer mix
# --------------------------------------------------------------------------- #
class MixFn(Function):
    """sum_l softmax(w)[l] * H[l]  (transformer_faces_objects.py:355-364)."""

    @staticmethod
    def forward(ctx, stack, w):
        L = stack.shape[0]
        n = stack[0].numel()
        out = torch.empty_like(stack[0])
        call('tell_mix_fwd', stack, w.detach(), L, n, out, hip.dt(stack))
        ctx.save_for_backward(stack)
        ctx.w = w
        return out

    @staticmethod
    def backward(ctx, dout):
        stack, = ctx.saved_tensors
        w = ctx.w
        L = stack.shape[0]
        n = stack[0].numel()
        nb = 1024                       # 4 workgroups per CU keep 25 x 16-byte loads per lane in flight: 159 us (5.5 TB/s); 512 blocks took 248
        partial = torch.empty(nb, L, dtype=torch.float32, device=stack.device)
        call('tell_mix_bwd', stack, dout.contiguous(), L, n, partial, nb, hip.dt(stack))
        gw = grad_buffer(w)
        if partial.is_cuda:         # column sums + the softmax chain rule over the 25 logits: one launch
            call('tell_mix_wgrad', partial, nb, L, w.detach().float().contiguous(), gw)
            return None, None
        dsm = partial.sum(0)                                                 # d loss / d softmax(w)
        sm = torch.softmax(w.detach().float(), dim=0)
        gw.add_(sm * (dsm - (sm * dsm).sum()))
        return None, None


def mix_layers(stack, w):
    return MixFn.apply(stack, w)


# --------------------------------------------------------------------------- #
# adaptive input embedding (+ sinusoidal positions)
# --------------------------------------------------------------------------- #
def partition_ids(ids_flat, cutoffs, pad_idx, want_slot=True, want_head_target=False):
    """Device-side band partition of int64 ids (replaces the mask/nonzero logic of
    adaptive.py:64-74 and softmax.py:144-167).  No host synchronisation."""
    import ctypes
    N = ids_flat.numel()
    nb = len(cutoffs)
    dev = ids_flat.device
    i32 = dict(dtype=torch.int32, device=dev)
    part = {
        'rows': torch.empty(nb, N, **i32), 'local': torch.empty(nb, N, **i32),
        'count': torch.empty(nb, **i32), 'slot': torch.empty(N, **i32) if want_slot else None,
        'head_target': torch.empty(N, **i32) if want_head_target else None,
        'n_valid': torch.empty(1, **i32), 'N': N,
    }
    cut = (ctypes.c_int * nb)(*[int(c) for c in cutoffs])
    call('tell_adaptive_partition', ids_flat, N, ctypes.cast(cut, ctypes.c_void_p).value, nb, int(pad_idx),
         part['rows'], part['local'], part['count'], part['slot'], part['head_target'], part['n_valid'])
    return part


class AdaptiveEmbedFn(Function):
    """scale * proj_band(emb_band[id - lo]) + sinusoid[position]  written directly in the
    decoder's T x B x C layout (adaptive.py:61-76, positional.py:167-211,
    sum_text_field_embedder.py:117-118)."""

    @staticmethod
    def forward(ctx, ids, pos_table, cutoffs, scale, pos_pad, start_pos, padding_idx, *tables):
        # tables = emb_0, proj_0, emb_1, proj_1, ...
        B, T = ids.shape
        N = B * T
        nb = len(cutoffs)
        dtype = rt.compute_dtype()
        E = tables[1].shape[0]
        ids_flat = ids.reshape(-1).contiguous()
        part = partition_ids(ids_flat, cutoffs, pad_idx=-1)
        band_out = torch.empty(nb * N, E, dtype=dtype, device=ids.device)
        rows_saved = []
        rows_all = zeros_group([((N, tables[2 * b].shape[1]), dtype) for b in range(nb)], ids.device)
        for b in range(nb):
            emb, proj = tables[2 * b], tables[2 * b + 1]
            dim = emb.shape[1]
            rows_b = rows_all[b]                                           # rows >= count must be 0, not garbage
            call('tell_gather_rows', weight(emb), dim, part['local'][b], part['count'][b:b + 1], N, rows_b, dim,
                 dim, hip.dt(dtype))
            gemm(rows_b, weight(proj), out=band_out[b * N:(b + 1) * N], m_dev=part['count'][b:b + 1])
            rows_saved.append(rows_b)
        out = torch.empty(T, B, E, dtype=dtype, device=ids.device)
        call('tell_embed_finalize', band_out, part['slot'], ids_flat, pos_table, pos_table.shape[0], out, B, T, E,
             float(scale), int(pos_pad), int(start_pos), 1, hip.dt(dtype))
        ctx.part, ctx.rows_saved, ctx.tables = part, rows_saved, tables
        ctx.meta = (B, T, E, nb, scale, padding_idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        part, rows_saved, tables = ctx.part, ctx.rows_saved, ctx.tables
        B, T, E, nb, scale, padding_idx = ctx.meta
        N = B * T
        dout = dout.contiguous()
        dband, = zeros_group([((nb * N, E), dout.dtype)], dout.device)
        call('tell_embed_finalize_bwd', dout, part['slot'], dband, B, T, E, float(scale), 1, hip.dt(dout))
        for b in range(nb):
            emb, proj = tables[2 * b], tables[2 * b + 1]
            dim = emb.shape[1]
            cnt = part['count'][b:b + 1]
            dy = dband[b * N:(b + 1) * N]                      # rows >= count are zero
            if proj.requires_grad:
                gemm_tn(dy, rows_saved[b], out=grad_buffer(proj), accumulate=True, k_dev=cnt)
            if emb.requires_grad:
                drows = gemm_nn(dy, weight(proj), b_t=lambda: weight_t(proj), m_dev=cnt)[:, :dim]
                if not drows.is_contiguous():
                    drows = drows.contiguous()
                call('tell_embed_table_grad', drows, drows.stride(0), part['local'][b], cnt, N,
                     grad_buffer(emb), dim, int(padding_idx), hip.dt(drows))
        return (None,) * (7 + len(tables))


def adaptive_embed(ids, pos_table, cutoffs, scale, pos_pad, start_pos, padding_idx, tables):
    return AdaptiveEmbedFn.apply(ids, pos_table, tuple(cutoffs), scale, pos_pad, start_pos, padding_idx,
                                 *tables)


# --------------------------------------------------------------------------- #
# adaptive softmax loss
# --------------------------------------------------------------------------- #
def _pad8(n):
    return _round_up(n, 8)


class AdaptiveLossFn(Function):
    """sum of per-cluster cross entropies (softmax.py:144-191 + adaptive_loss.py:27-73),
    returned in nats together with sample_size as DEVICE scalars.  Static shapes:
    tail rows are compacted on the device, tail GEMMs run with a device row count."""

    @staticmethod
    def forward(ctx, x, target, cutoffs, pad_idx, emb0, class_proj, *tails):
        # tails = proj_0, emb_1, proj_1, emb_2, ...   (tail i: logits = emb_{i+1} (proj_i x))
        N, E = x.shape[0] * x.shape[1], x.shape[2]
        dtype = x.dtype
        dev = x.device
        x2 = as2dc(x)
        n_tails = len(tails) // 2
        c0 = cutoffs[0]
        tflat = target.reshape(-1).contiguous()
        part = partition_ids(tflat, cutoffs, pad_idx=pad_idx, want_slot=False, want_head_target=True)
        # ---- head: [emb0 ; class_proj] x
        n_head = c0 + n_tails
        w_head = torch.empty(n_head, E, dtype=dtype, device=dev)
        if w_head.is_cuda:
            call('tell_cast', weight(emb0), hip.dt(dtype), w_head, hip.dt(dtype), c0 * E)
            call('tell_cast', weight(class_proj), hip.dt(dtype), w_head[c0:], hip.dt(dtype), (n_head - c0) * E)
        else:
            w_head[:c0] = weight(emb0)
            w_head[c0:] = weight(class_proj)
        # (rows of every fp32 logits buffer start on 16 bytes: an odd row length - 5002, 30265 - would make every store
        #  of the GEMM epilogue a 4-byte one: the 124 MB of tail logits took 137 us that way, 28 us as whole row pieces)
        head_logits = torch.empty(N, _round_up(n_head, 4), dtype=torch.float32, device=dev)[:, :n_head]
        gemm(x2, w_head, out=head_logits)
        lse_h = torch.empty(N, dtype=torch.float32, device=dev)
        loss_rows = torch.empty(N, dtype=torch.float32, device=dev)
        call('tell_ce_fwd', head_logits, head_logits.stride(0), N, n_head, part['head_target'], None, None,
             int(pad_idx), lse_h, loss_rows)
        zs = zeros_group([((1,), torch.float32)] +
                         [spec for i in range(n_tails) for spec in (((N, E), dtype), ((N, tails[2 * i].shape[0]), dtype))],
                         dev)
        total = zs[0]
        call('tell_sum_f32', loss_rows, N, None, total, 1)
        saved_tails = []
        for i in range(n_tails):
            proj, emb = tails[2 * i], tails[2 * i + 1]
            band = i + 1
            cnt = part['count'][band:band + 1]
            xg = zs[1 + 2 * i]
            call('tell_gather_rows', x2, E, part['rows'][band], cnt, N, xg, E, E, hip.dt(dtype))
            h = zs[2 + 2 * i]
            gemm(xg, weight(proj), out=h, m_dev=cnt)
            V = emb.shape[0]
            logits = torch.empty(N, _round_up(V, 4), dtype=torch.float32, device=dev)[:, :V]
            gemm(h, weight(emb), out=logits, m_dev=cnt)
            lse_t = torch.empty(N, dtype=torch.float32, device=dev)
            lrow = torch.empty(N, dtype=torch.float32, device=dev)
            # target of compacted row j is local[band][j]; quirk: ignore_index also applies here
            call('tell_ce_fwd', logits, logits.stride(0), N, V, part['local'][band], None, cnt, int(pad_idx),
                 lse_t, lrow)
            call('tell_sum_f32', lrow, N, cnt, total, 1)
            saved_tails.append((xg, h, logits, lse_t))
        ctx.saved = (x2, w_head, head_logits, lse_h, part, saved_tails)
        ctx.params = (emb0, class_proj, tails, cutoffs, pad_idx)
        ctx.xshape = x.shape
        ctx.mark_non_differentiable(part['n_valid'])
        ctx.set_materialize_grads(False)
        return total, part['n_valid']

    @staticmethod
    def backward(ctx, gtotal, _gn):
        x2, w_head, head_logits, lse_h, part, saved_tails = ctx.saved
        emb0, class_proj, tails, cutoffs, pad_idx = ctx.params
        if gtotal is None:
            return (None,) * (6 + len(tails))
        N, E = x2.shape
        dtype, dev = x2.dtype, x2.device
        c0 = cutoffs[0]
        n_tails = len(tails) // 2
        n_head = c0 + n_tails
        gscale = gtotal.reshape(1).float().contiguous()
        # ---- head
        zs = zeros_group([((N, _round_up(n_head, _vec(dtype))), dtype)] +
                         [spec for i in range(n_tails) for spec in
                          (((N, _round_up(tails[2 * i + 1].shape[0], _vec(dtype))), dtype), ((N, tails[2 * i].shape[0]), dtype))],
                         dev)
        dl = zs[0]
        call('tell_ce_bwd', head_logits, head_logits.stride(0), N, n_head, part['head_target'], None, None,
             int(pad_idx), lse_h, gscale, dl, dl.stride(0), hip.dt(dtype))
        dx = gemm_nn(dl, w_head)                                  # [N, E]; dl's padding columns are zero
        if not dx.is_contiguous():
            dx = dx.contiguous()
        if emb0.requires_grad:
            gemm_tn(dl[:, :c0], x2, out=grad_buffer(emb0), accumulate=True)
        if class_proj.requires_grad:
            gemm_tn(dl[:, c0:n_head], x2, out=grad_buffer(class_proj), accumulate=True)
        # ---- tails
        for i in range(n_tails):
            proj, emb = tails[2 * i], tails[2 * i + 1]
            band = i + 1
            cnt = part['count'][band:band + 1]
            xg, h, logits, lse_t = saved_tails[i]
            V = emb.shape[0]
            dlt = zs[1 + 2 * i]
            call('tell_ce_bwd', logits, logits.stride(0), N, V, part['local'][band], None, cnt, int(pad_idx),
                 lse_t, gscale, dlt, dlt.stride(0), hip.dt(dtype))
            dh = zs[2 + 2 * i]
            gemm_nn(dlt, weight(emb), b_t=lambda emb=emb: weight_t(emb), out=dh, m_dev=cnt, zero_rows=True)
            if emb.requires_grad:                                  # rows >= count of dlt are zero: not read (k_dev)
                gemm_tn(dlt[:, :V], h, out=grad_buffer(emb), accumulate=True, k_dev=cnt)
            if proj.requires_grad:
                gemm_tn(dh, xg, out=grad_buffer(proj), accumulate=True, k_dev=cnt)
            dxg = gemm_nn(dh, weight(proj), b_t=lambda proj=proj: weight_t(proj), m_dev=cnt)
            if not dxg.is_contiguous():
                dxg = dxg.contiguous()
            call('tell_scatter_add_rows', dxg, dxg.stride(0), part['rows'][band], cnt, N, dx, dx.stride(0), E,
                 hip.dt(dtype))
        return (dx.view(ctx.xshape),) + (None,) * (5 + len(tails))


class LossBitsFn(Function):
    """sum of nats -> bits per target token (transformer_faces_objects.py:85-88): loss = total / ln 2 / n, one launch
    each way instead of the five elementwise ones of the expression."""

    @staticmethod
    def forward(ctx, total, n_valid):
        out = torch.empty(1, dtype=torch.float32, device=total.device)
        call('tell_loss_bits', total, n_valid, out)
        ctx.n = n_valid
        return out

    @staticmethod
    def backward(ctx, g):
        gt = torch.empty(1, dtype=torch.float32, device=g.device)
        call('tell_loss_bits', g.reshape(1).float().contiguous(), ctx.n, gt)
        return gt, None


def loss_bits(total, n_valid):
    if total.is_cuda and n_valid.dtype == torch.int32:
        return LossBitsFn.apply(total.reshape(1), n_valid.reshape(1)).reshape(())
    return (total / math.log(2) / n_valid.to(torch.float32)).reshape(())


def adaptive_loss(x, target, cutoffs, pad_idx, emb0, class_proj, tails):
    """-> (loss_sum_nats [1] fp32 device tensor, n_valid [1] int32 device tensor)."""
    return AdaptiveLossFn.apply(x, target, tuple(cutoffs), pad_idx, emb0, class_proj, *tails)


def adaptive_log_probs(x2, cutoffs, emb0, class_proj, tails, want_full=False, topk=0):
    """Generation head (softmax.py:193-222 + topk(1)): -> (token int32 [N], logprob fp32 [N], full or None);
    topk = k > 0: -> (tokens int32 [N,k], logprobs fp32 [N,k], None), best first (beam search)."""
    N, E = x2.shape
    dev = x2.device
    c0 = cutoffs[0]
    n_tails = len(tails) // 2
    from . import decode
    wide = decode.HEAD_COMPOSED and all(tails[2 * i].shape[1] == E for i in range(n_tails))    # (head_step's one-product form)
    if (not want_full and decode.ENABLED and N <= (decode.MAX_ROWS_WIDE if wide else decode.MAX_ROWS) and
            x2.dtype == torch.bfloat16 and
            E % 1024 == 0 and 1 <= n_tails <= 3 and all(tails[2 * i].shape[0] % 8 == 0 for i in range(n_tails))):
        return decode.head_step(x2, cutoffs, emb0, class_proj, tails, topk)
    w_head = _cached(emb0, ('whead', class_proj._version, class_proj.data_ptr()), lambda: torch.cat(
        [weight(emb0), weight(class_proj)], dim=0).contiguous())
    def logits(a, w):                    # fp32 rows start on 16 bytes (see AdaptiveLossFn): vector stores in the epilogue
        out = torch.empty(a.shape[0], _round_up(w.shape[0], 4), dtype=torch.float32, device=dev)[:, :w.shape[0]]
        return gemm(a, w, out=out)
    head = logits(x2, w_head)
    tl, ld, nn_ = [None] * 3, [0] * 3, [0] * 3
    for i in range(n_tails):
        proj, emb = tails[2 * i], tails[2 * i + 1]
        h = gemm(x2, weight(proj))
        tl[i] = logits(h, weight(emb))
        ld[i], nn_[i] = tl[i].stride(0), tl[i].shape[1]
    vocab = c0 + sum(nn_)
    if topk:
        tokens = torch.empty(N, topk, dtype=torch.int32, device=dev)
        lps = torch.empty(N, topk, dtype=torch.float32, device=dev)
        call('tell_adaptive_logprob_topk', head, head.stride(0), c0, n_tails, tl[0], ld[0], nn_[0], tl[1], ld[1],
             nn_[1], tl[2], ld[2], nn_[2], N, int(topk), tokens, lps)
        return tokens, lps, None
    full = torch.empty(N, vocab, dtype=torch.float32, device=dev) if want_full else None
    token = torch.empty(N, dtype=torch.int32, device=dev)
    token_lp = torch.empty(N, dtype=torch.float32, device=dev)
    call('tell_adaptive_logprob_argmax', head, head.stride(0), c0, n_tails, tl[0], ld[0], nn_[0], tl[1], ld[1],
         nn_[1], tl[2], ld[2], nn_[2], N, full, vocab if want_full else 0, token, token_lp)
    return token, token_lp, full


LN2 = math.log(2.0)
