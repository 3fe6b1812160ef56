"""One generation step of the DynamicConv decoders (decoder_faces_objects.py:224-352 / decoder_flattened.py at T = 1,
driven by transformer_faces_objects.py:443-494) as 12 launches per layer - csrc/decode.hip.

The training kernels treat a [1, M, E] step as a tiny GEMM problem; here every linear layer streams its weights once
with all CUs (tell_skinny_linear), the producer of a LayerNorm input leaves `residual + branch` in fp32 and one small
launch normalises it for the consumers, the DynamicConv step is one kernel (tell_dynconv_step) and the 2 / 4 context
attentions are one launch against the projected K/V cache (tell_attn_decode).  Same arithmetic as the layer-by-layer
path up to bf16 rounding points (pre-norm sums stay fp32 here); tests/test_gpu_fullsize.py bounds the difference.
"""
import ctypes
import os
import weakref

import torch

from . import ops
from . import hip as hip_mod
from .hip import call

ENABLED = os.environ.get('TELL_DECODE_FUSED', '1') != '0'       # A/B aid: 0 = the layer-by-layer step
# rows (batch x beam) up to which the weight-streaming step is used (round 3, at 32 and 128 rows: 0.51 / 1.07 ms per step
# against 0.83 / 1.29 ms layer by layer); larger batches are GEMM-shaped again.  Round 6 (operands staged through LDS,
# 32-row workgroups for any row count): 256 rows 804 -> 752 us (beam 4 at 64 captions per batch), 1169 -> 1081 us (256 greedy
# rows); 512 rows 1042 -> 1266 us - the limit is 256 now.
MAX_ROWS = int(os.environ.get('TELL_DECODE_ROWS', '256'))


# LayerNorms of the step FOLDED into their consumers (round 5): the producer of a pre-norm `residual + branch` leaves it in
# fp32 (the residual of the next sub-layer is rebuilt from it) AND in bf16; the consuming linear runs on the bf16 raw rows
# with weights pre-scaled by gamma, gathers the row statistics from its own MFMA operands and corrects in its epilogue
# (csrc/decode.hip FOLD): 12 LayerNorm launches per step become one (the head's input).  TELL_DECODE_FOLD=0: round-4 form.
FOLD = os.environ.get('TELL_DECODE_FOLD', '1') != '0'
HEAD_GROUPED = os.environ.get('TELL_HEAD_GROUPED', '1') != '0'      # A/B aid: the two tail-table products as one launch
# the generation loop's projected K / V cache HEAD-MAJOR ([B, H, S, 64]: models/transformer.py _decode_stepper); 0 = [S, B, E]
KV_HEAD_MAJOR = os.environ.get('TELL_KV_HEAD_MAJOR', '1') != '0'
# ... and, on top of it, PACKED for the matrix cores (PackedKV below; tell_attn_decode_packed); 0 = the VALU kernel on the
# head-major cache
KV_PACKED = os.environ.get('TELL_KV_PACKED', '1') != '0'
ENABLED_LAYER_PACKED = os.environ.get('TELL_KV_PACKED_LAYERS', '1') != '0'
PACKED_MIN_HYP = int(os.environ.get('TELL_PACKED_MIN_HYP', '2'))   # hypotheses per sample from which the packed cache is taken
# the per-token bookkeeping launch (tell_greedy_update / tell_beam_update) as the LAST launch of the captured step: the host's
# part of a decode step is one graph replay.  0 = a host-side launch behind every replay (A/B aid)
IN_GRAPH_BOOK = os.environ.get('TELL_DECODE_BOOK_IN_GRAPH', '1') != '0'
# the step's token embedding as a lookup in a pre-projected [V, E] fp32 table (embed_step); 0 = gather + skinny linear per step
EMBED_TABLE = os.environ.get('TELL_DECODE_EMBED_TABLE', '1') != '0'
# the adaptive-softmax head of the step as ONE product against [emb_0; class_proj; table_1 . proj_1; table_2 . proj_2] (head_step)
HEAD_COMPOSED = os.environ.get('TELL_HEAD_COMPOSED', '1') != '0'
# the two precomputed forms (embedding lookup, composed head) do not care how many rows a step has: also on the layer-by-layer
# path above MAX_ROWS (beam 4 at 128 captions per batch = 512 rows)
MAX_ROWS_WIDE = 1024
HEAD_TILE128 = os.environ.get('TELL_HEAD_TILE128', '1') != '0'


def _folded(w_param_key, w, lns, seg):
    """(W' bf16 [N, K], s fp32 [K / seg][N], c fp32 [N]) of `LN_seg(x) . W^T` for consecutive column segments of width seg:
    W'[n][k] = W[n][k] gamma_seg[k], s[seg][n] = sum_k W'[n][k] over the segment (of the ROUNDED W': what the matrix cores
    multiply), c[n] = sum_k W[n][k] beta[k].  Cached with the weight's cache entry (dropped when the weights change)."""
    def make():
        wf = w.float()
        K = wf.shape[1]
        g = torch.cat([ln.weight.detach().float() for ln in lns])
        b = torch.cat([ln.bias.detach().float() for ln in lns])
        assert g.numel() == K and K % seg == 0
        wp = (wf * g[None, :]).to(torch.bfloat16).contiguous()
        s_vec = wp.float().view(wf.shape[0], K // seg, seg).sum(2).t().contiguous()          # [K / seg][N]
        c_vec = (wf @ b).contiguous()
        return wp, s_vec, c_vec
    # ONE entry per (weight parameter, consumer): overwritten in place when it goes stale.  (Round 5 keyed it on w.data_ptr()
    # too - the working weight is re-allocated with every weights epoch, so every train-then-validate round left a dead
    # gamma-scaled copy of every decoder linear behind.)  Stale = the weight's own stamp (ops._fresh: weights epoch, dtype,
    # version, address) or the guard below: the LayerNorms' torch versions and - the optimizer kernel bypasses those - the
    # weights epoch when a LayerNorm is trainable.
    trainable = any(getattr(t, 'requires_grad', False) for ln in lns for t in (ln.weight, ln.bias))
    guard = (ops.rt.weights_epoch() if trainable else -1,
             tuple((ln.weight._version, ln.bias._version, ln.weight.data_ptr(), ln.bias.data_ptr()) for ln in lns))
    key = ('ln_fold', seg, len(lns))
    hit = ops._fresh(w_param_key, key)
    if hit is not None and hit[1][0] == guard:
        return hit[1][1]
    val = make()
    ops._wcache[(id(w_param_key), key)] = (ops._stamp(w_param_key), (guard, val), weakref.ref(w_param_key))
    return val


def _ptrs(items):
    return (ctypes.c_void_p * len(items))(*[(t.data_ptr() if torch.is_tensor(t) else (t or 0)) for t in items])


def _longs(vals):
    return (ctypes.c_long * len(vals))(*vals)


def _ints(vals):
    return (ctypes.c_int * len(vals))(*vals)


def usable(dec, X, incremental_state, kv_cache):
    """The fused step covers what the expt/ configs build: post-LN layers, GLU + DynamicConv with 64-wide heads,
    bf16, eval, a static (fixed-shape) incremental state and projected K/V."""
    if not (ENABLED and kv_cache is not None and incremental_state is not None and incremental_state.get('_static') and
            incremental_state.get('_ring')):
        return False
    if dec.training or not X.is_cuda or X.dtype != torch.bfloat16 or ops.rt.compute_dtype() != torch.bfloat16:
        return False
    if X.shape[0] != 1 or X.shape[1] > MAX_ROWS or getattr(dec, 'normalize', False):
        return False
    E = X.shape[2]
    for layer in dec.layers:
        conv = layer.conv
        if (layer.normalize_before or layer.need_attn or not layer.glu or type(conv).__name__ != 'DynamicConv1dTBC' or
                layer.conv_dim != E or E % 512 or conv.num_heads * 64 != E or not conv.ring_usable() or
                conv.weight_linear.bias is not None or not 1 <= len(layer.context_names) <= 4 or
                layer.fc1.out_features % 256):
            return False
        for name in layer.context_names:
            m = layer.context_attns[name]
            ent = kv_cache[0][name]
            n_keys = ent.shape[0] if isinstance(ent, PackedKV) else ent[0].shape[0]
            if m.head_dim != 64 or m.bias_k is None or not m.add_zero_attn or n_keys > 2048:
                return False
    return True


_SPLIT_WS = {}
CUR_LANE = [0]          # which decode lane is being issued / recorded (models/transformer.py): lanes own their split workspace


def split_workspace(device):
    """Workspace of the skinny linears that share a reduction between workgroups (tell_skinny_linear split_ws): 64 KB of
    arrival counters - zeroed HERE, once; the kernels keep them consistent - and 16 MB for partial tiles.  One per device:
    the launches that use it (the decode steps of this process) are ordered with respect to each other - one stream, or
    replays of graphs captured from it; a caller that decodes on two streams at once gives each its own (pass split_ws to
    tell_skinny_linear directly).  Created outside any stream capture (the eager warm step of a decode loop comes first)."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), CUR_LANE[0])
    ws = _SPLIT_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None                                   # (never allocate + zero inside a capture: no split for this graph)
        ws = _SPLIT_WS[key] = torch.zeros((65536 + (16 << 20)) // 4, dtype=torch.int32, device=device)
    return ws


def _skinny(ins, ld_in, ws, biases, outs, ld_out, M, N, K, pro=0, gammas=None, betas=None, seg=0, eps=1e-5,
            stats_out=None, act=0, scale=1.0, res=None, ld_res=0, res_raw=None, res_stats=None, res_ln=None,
            res_f32=None, out2=None, out2_from=0, out_f32=False):
    n = len(ins)
    work = None
    if pro in (1, 2):                                                    # LayerNorm rows as their own launch
        work = torch.empty(M, K, dtype=torch.bfloat16, device=ws[0].device)
    sws = split_workspace(ws[0].device) if (n == 1 and M <= 32 and K >= 2048 and N <= 1536) else None
    call('tell_skinny_linear', n, _ptrs(ins), ld_in, pro, _ptrs(gammas) if gammas else None,
         _ptrs(betas) if betas else None, seg, eps, stats_out, work, _ptrs(ws), ws[0].stride(0),
         _ptrs(biases) if biases is not None else None, act, scale, res, ld_res, res_raw,
         res_raw.stride(0) if res_raw is not None else 0, res_stats,
         res_ln.weight if res_ln is not None else None, res_ln.bias if res_ln is not None else None,
         res_f32, res_f32.stride(0) if res_f32 is not None else 0, out2,
         out2.stride(0) if out2 is not None else 0, out2_from, _ptrs(outs), ld_out, 1 if out_f32 else 0, M, N, K,
         sws, sws.numel() * 4 if sws is not None else 0)


def embed_usable(embedder, ids, incremental_state):
    ad = embedder.token_embedder_adaptive
    return (ENABLED and incremental_state is not None and incremental_state.get('_static') and not embedder.training and
            ids.is_cuda and ids.shape[1] == 1 and ids.shape[0] <= (MAX_ROWS_WIDE if EMBED_TABLE else MAX_ROWS) and
            ops.rt.compute_dtype() == torch.bfloat16 and len(ad.cutoff) <= 4 and ad.embed_size % 1024 == 0 and
            all(s[0].weight.shape[1] % 8 == 0 for s in ad.embeddings))


def embed_step(embedder, ids, start):
    """scale * proj_band(table_band[id]) + sinusoid[position] of one token per row (sum_text_field_embedder.py:117-118 at
    T = 1) as two launches: the table rows placed side by side by band, then ONE skinny linear against
    [proj_0 | proj_1 | ..] with the scale and the sinusoid row in its epilogue.  -> [1, M, E] bf16."""
    ad, po = embedder.token_embedder_adaptive, embedder.token_embedder_position
    M, E = ids.shape[0], ad.embed_size
    if EMBED_TABLE and E % 4 == 0:
        # generation: every token's projected, scaled embedding ONCE ([V, E] fp32: 206 MB at V = 50265 - what 288 GB are for),
        # rebuilt when the weights change (ops._cached: weights epoch / versions); the step's embedding is then one lookup launch
        prm = [t for s in ad.embeddings for t in (s[0].weight, s[1].weight)]

        def full_table():
            rows = [ops.gemm(ops.weight(s[0].weight), ops.weight(s[1].weight), out_dtype=torch.float32,
                             alpha=float(ad.embed_scale)) for s in ad.embeddings]
            return torch.cat(rows, 0).contiguous()
        table = ops._cached(prm[0], ('embed_full_table',) + tuple((p._version, p.data_ptr()) for p in prm[1:]), full_table)
        out = torch.empty(M, E, dtype=torch.bfloat16, device=ids.device)
        call('tell_embed_lookup_step', ids.reshape(M), M, table, table.shape[0], po.weights, po.weights.shape[0],
             int(po.padding_idx), int(start), out, E)
        return out.view(1, M, E)
    dims = [s[0].weight.shape[1] for s in ad.embeddings]
    offs = [sum(dims[:i]) for i in range(len(dims))]
    ktot = -(-sum(dims) // 1024) * 1024
    lo = [0] + list(ad.cutoff[:-1])
    projs = [s[1].weight for s in ad.embeddings]

    def make():
        w = torch.zeros(E, ktot, dtype=torch.bfloat16, device=ids.device)
        for o, d, p in zip(offs, dims, projs):
            w[:, o:o + d] = ops.weight(p)
        return w
    w_cat = ops._cached(projs[0], ('embed_cat',) + tuple((p._version, p.data_ptr()) for p in projs[1:]), make)
    cat = torch.empty(M, ktot, dtype=torch.bfloat16, device=ids.device)
    pos = torch.empty(M, E, dtype=torch.float32, device=ids.device)
    call('tell_embed_gather_step', ids.reshape(M), M, len(dims), _ptrs([ops.weight(s[0].weight) for s in ad.embeddings]),
         _ints(lo), _ints(list(ad.cutoff)), _ints(dims), _ints(offs), cat, ktot, po.weights, po.weights.shape[0],
         int(po.padding_idx), int(start), pos, E)
    out = torch.empty(M, E, dtype=torch.bfloat16, device=ids.device)
    _skinny([cat], ktot, [w_cat], None, [out], E, M, E, ktot, scale=float(ad.embed_scale), res_f32=pos)
    return out.view(1, M, E)


def head_step(x2, cutoffs, emb0, class_proj, tails, topk=0):
    """Greedy head of a generation step (softmax.py:193-222 + topk(1)) as four launches: head logits | cluster logits |
    the tails' projected inputs from ONE skinny linear over [emb_0; class_proj; proj_1; proj_2] (logits fp32, the
    projections once more in bf16), one skinny linear per tail table, the register-resident arg-max.
    -> (token int32 [N], log-prob fp32 [N], None); topk = k > 0 (beam search): the k best of every row, best first,
    (tokens int32 [N,k], log-probs fp32 [N,k], None)."""
    N, E = x2.shape
    dev = x2.device
    c0, n_tails = cutoffs[0], len(tails) // 2
    projs = [tails[2 * i] for i in range(n_tails)]
    if HEAD_COMPOSED and 1 <= n_tails <= 3 and all(p.shape[1] == E for p in projs):
        # Round 6: during generation the weights do not move, so a tail's projection and its table are ONE matrix,
        # W_i = table_i . proj_i ([n_i, E]; softmax.py:207-214 computes (x . proj_i^T) . table_i^T = x . W_i^T), composed once per
        # weights state in fp32 and rounded to bf16 like every other working weight.  Head rows, cluster rows and the composed
        # tails are stacked into one [V + n_tails (+ padding to 16-byte segments), E] matrix: the head is ONE product (the
        # grouped GEMM's 64x64 fp32 tiles, which stream 100 MB at 4 TB/s where the skinny form took 13 us for 14 MB) + the
        # register-resident arg-max / top-k, instead of skinny linear + grouped GEMM + arg-max.
        embs = [tails[2 * i + 1] for i in range(n_tails)]
        prm = [emb0, class_proj] + projs + embs

        def make_big():
            segs, offs, off = [], [], 0

            def push(t):
                nonlocal off
                offs.append(off)
                pad = -t.shape[0] % 4
                segs.append(t if not pad else torch.cat([t, t.new_zeros(pad, t.shape[1])], 0))
                off += t.shape[0] + pad
            push(torch.cat([ops.weight(emb0), ops.weight(class_proj)], 0))
            for p_, e_ in zip(projs, embs):
                push(ops.gemm(ops.weight(e_), ops.weight(p_).t().contiguous(), out_dtype=torch.float32).to(torch.bfloat16))
            return torch.cat(segs, 0).contiguous(), offs
        w_big, offs = ops._cached(emb0, ('whead_composed',) + tuple((p._version, p.data_ptr()) for p in prm[1:]), make_big)
        LD = w_big.shape[0]
        logits = torch.empty(N, LD, dtype=torch.float32, device=dev)
        if N >= 128 and HEAD_TILE128:
            # 128 rows and more: 128-row tiles, so that the 100 MB table is streamed once, not once per 64-row tile (the
            # launcher's own rule keeps 64x64 tiles below 512 tiles of 128x128; the choice is recorded with the captured launch)
            with hip_mod.options(group_tile=128):
                ops.gemm_grouped([dict(a=x2, b=w_big, out=logits, form='nt')])
        else:
            ops.gemm_grouped([dict(a=x2, b=w_big, out=logits, form='nt')])
        ns = [e.shape[0] for e in embs] + [0] * (3 - n_tails)
        tl = [logits[:, offs[1 + i]:] if i < n_tails else None for i in range(3)]
        lds = [LD if i < n_tails else 0 for i in range(3)]
        if topk:
            tokens = torch.empty(N, topk, dtype=torch.int32, device=dev)
            lps = torch.empty(N, topk, dtype=torch.float32, device=dev)
            call('tell_adaptive_logprob_topk', logits, LD, c0, n_tails, tl[0], lds[0], ns[0], tl[1], lds[1], ns[1], tl[2], lds[2],
                 ns[2], N, int(topk), tokens, lps)
            return tokens, lps, None
        token = torch.empty(N, dtype=torch.int32, device=dev)
        token_lp = torch.empty(N, dtype=torch.float32, device=dev)
        call('tell_adaptive_logprob_argmax', logits, LD, c0, n_tails, tl[0], lds[0], ns[0], tl[1], lds[1], ns[1], tl[2], lds[2],
             ns[2], N, None, 0, token, token_lp)
        return token, token_lp, None
    w_all = ops._cached(emb0, ('whead_all', class_proj._version, class_proj.data_ptr()) +
                        tuple((p._version, p.data_ptr()) for p in projs),
                        lambda: torch.cat([ops.weight(emb0), ops.weight(class_proj)] + [ops.weight(p) for p in projs],
                                          dim=0).contiguous())
    n_head = c0 + n_tails
    n_all = w_all.shape[0]
    ld = -(-n_all // 4) * 4
    head = torch.empty(N, ld, dtype=torch.float32, device=dev)
    hdims = [p.shape[0] for p in projs]
    h = torch.empty(N, sum(hdims), dtype=torch.bfloat16, device=dev)
    _skinny([x2], x2.stride(0), [w_all], None, [head], ld, N, n_all, E, out2=h, out2_from=n_head, out_f32=True)
    tl, lds, ns = [None] * 3, [0] * 3, [0] * 3
    off = 0
    big = []                                                 # the large tail tables: ONE grouped launch (round 5)
    for i in range(n_tails):
        emb = ops.weight(tails[2 * i + 1])
        n_i = emb.shape[0]
        lds[i], ns[i] = -(-n_i // 4) * 4, n_i
        tl[i] = torch.empty(N, lds[i], dtype=torch.float32, device=dev)
        if n_i * hdims[i] <= 4096 * 1024 and hdims[i] % 256 == 0:
            _skinny([h[:, off:off + hdims[i]]], h.stride(0), [emb], None, [tl[i]], lds[i], N, n_i, hdims[i], out_f32=True)
        elif HEAD_GROUPED and hdims[i] % 64 == 0 and off % 8 == 0:
            big.append(dict(a=h[:, off:off + hdims[i]], b=emb, out=tl[i][:, :n_i], form='nt'))
        else:       # tens of MB of table: the MFMA GEMM's 64-column tiles re-read the rows 4x less often per weight byte
            ops.gemm(h[:, off:off + hdims[i]], emb, out=tl[i][:, :n_i])
        off += hdims[i]
    if len(big) == 1:
        ops.gemm(big[0]['a'], big[0]['b'], out=big[0]['out'])
    elif big:       # two dependent launches of 235 / 473 column tiles (12 + 16 us) -> one launch of 708
        ops.gemm_grouped(big)
    if topk:
        tokens = torch.empty(N, topk, dtype=torch.int32, device=dev)
        lps = torch.empty(N, topk, dtype=torch.float32, device=dev)
        call('tell_adaptive_logprob_topk', head, ld, c0, n_tails, tl[0], lds[0], ns[0], tl[1], lds[1], ns[1], tl[2], lds[2],
             ns[2], N, int(topk), tokens, lps)
        return tokens, lps, None
    token = torch.empty(N, dtype=torch.int32, device=dev)
    token_lp = torch.empty(N, dtype=torch.float32, device=dev)
    call('tell_adaptive_logprob_argmax', head, ld, c0, n_tails, tl[0], lds[0], ns[0], tl[1], lds[1], ns[1], tl[2], lds[2],
         ns[2], N, None, 0, token, token_lp)
    return token, token_lp, None


class PackedKV:
    """Projected K / V of one (layer, context) in the layout tell_attn_decode_packed reads (csrc/decode.hip): the bias_k / bias_v
    row and the zero row are keys S and S + 1, the key axis is padded to Sp % 32 == 0, and both operands sit in MFMA FRAGMENT
    ORDER - a wave-wide 16-byte load reads one contiguous KB: kc [Bc, H, Sp / 16 key tiles, 2 halves of the head width, 64
    lanes, 8] (lane l = key l & 15 of the tile, elements half * 32 + (l >> 4) * 8 ..), vt [Bc, H, Sp / 32 blocks, 4 row tiles
    of the head width, 64 lanes, 8] (lane l = dimension tile * 16 + (l & 15), the block's keys 4 g + r | 16 + 4 g + r of k-group
    g = l >> 4 - the order the probabilities leave the score tiles in); mask [Bc, Sp] uint8 (1 = masked).  Built once per
    stepper (models/transformer.py), refilled per caption batch by `fill` (one permuting copy per operand and caption batch -
    against a hundred decode steps that read it)."""

    def __init__(self, mod, S, Bc, device):
        H = mod.num_heads
        self.S, self.Bc, self.H = int(S), int(Bc), H
        self.Sp = -(-(self.S + 2) // 32) * 32
        bf = dict(dtype=torch.bfloat16, device=device)
        self.knat = torch.zeros(Bc, H, self.Sp, 64, **bf)               # staging: keys / values in key order
        self.vnat = torch.zeros(Bc, H, self.Sp, 64, **bf)
        self.kc = torch.zeros(Bc, H, self.Sp * 64, **bf)
        self.vt = torch.zeros(Bc, H, 64 * self.Sp, **bf)
        self.mask = torch.ones(Bc, self.Sp, dtype=torch.uint8, device=device)
        self.mask[:, self.S:self.S + 2] = 0                             # the two virtual keys are never masked
        self.knat[:, :, self.S] = ops._bias_row(mod.bias_k, torch.bfloat16).view(H, 64)
        self.vnat[:, :, self.S] = ops._bias_row(mod.bias_v, torch.bfloat16).view(H, 64)
        self.shape = (self.S, Bc, H * 64)                               # (what the [S, B, E] tensors it replaces answer)

    def fill(self, k, v, mask):
        S, Bc, H, nb = self.S, self.Bc, self.H, self.Sp // 32
        if S:
            self.knat[:, :, :S].copy_(k.view(S, Bc, H, 64).permute(1, 2, 0, 3))
            self.vnat[:, :, :S].copy_(v.view(S, Bc, H, 64).permute(1, 2, 0, 3))
            if mask is not None:
                self.mask[:, :S].copy_(mask)
            else:
                self.mask[:, :S].zero_()
        # keys: [tile, key lr, half c, k-group lg, 8] -> [tile, c, lg, lr, 8]
        self.kc.view(Bc, H, 2 * nb, 2, 4, 16, 8).copy_(self.knat.view(Bc, H, 2 * nb, 16, 2, 4, 8).permute(0, 1, 2, 4, 5, 3, 6))
        # values: key (block, half, g, r) x dimension (tile rt, lr) -> [block, rt, g, lr, (half, r)]
        self.vt.view(Bc, H, nb, 4, 4, 16, 2, 4).copy_(self.vnat.view(Bc, H, nb, 2, 4, 4, 4, 16).permute(0, 1, 2, 6, 4, 7, 3, 5))


def layer_path_takes_packed(dec):
    """More rows than the weight-streaming step takes (MAX_ROWS): the layer-by-layer step hands the n context attentions of a
    layer to attn_decode_all - and with it to a PackedKV cache - exactly when DynamicConvDecoderLayer.forward groups them:
    post-LN layers with more than one context, no attention-weight export, bf16, the learned bias row + zero row, 64-wide heads."""
    if dec.training or ops.rt.compute_dtype() != torch.bfloat16 or not ENABLED_LAYER_PACKED:
        return False
    for layer in dec.layers:
        if layer.normalize_before or layer.need_attn or not 2 <= len(layer.context_names) <= 4:
            return False
        for name in layer.context_names:
            m = layer.context_attns[name]
            if m.head_dim != 64 or m.bias_k is None or not m.add_zero_attn:
                return False
    return True


def attn_decode_usable(mods, kv_layer, names, q):
    """tell_attn_decode takes: bf16, one query position per row, 64-wide heads, the learned bias row + the zero row,
    at most 4 contexts of at most 2048 cached keys."""
    if not (q.is_cuda and q.dtype == torch.bfloat16 and 1 <= len(mods) <= 4):
        return False
    if all(isinstance(kv_layer[nm], PackedKV) for nm in names):
        return all(m.head_dim == 64 for m in mods)
    for m, nm in zip(mods, names):
        k = kv_layer[nm][0]
        if m.head_dim != 64 or m.bias_k is None or not m.add_zero_attn or k.shape[0] > 2048 or k.dtype != torch.bfloat16:
            return False
        if k.shape[0] > 0 and (k.stride(-1) != 1 or kv_layer[nm][1].stride(-1) != 1 or q.shape[-2] % k.shape[1] != 0):
            return False
        if k.dim() == 4 and (k.shape[3] != 64 or kv_layer[nm][1].dim() != 4):      # head-major cache [S, B, H, 64] views
            return False
    return True


def attn_decode_all(mods, names, qs, kv_layer, contexts, M, E):
    """The n one-query context attentions of a layer against the projected K / V cache as ONE launch (multi_head.py:376-475
    at Tq = 1): qs[i] [M, E] bf16 (projected, scaled) -> [n, M, E] bf16.  The `beams` hypotheses of a sample (rows
    b * beams + j) share its cache: beams = M // (cached batch)."""
    n = len(mods)
    dev = qs[0].device
    a_all = torch.empty(n, M, E, dtype=torch.bfloat16, device=dev)
    if isinstance(kv_layer[names[0]], PackedKV):
        pk = [kv_layer[nm] for nm in names]
        call('tell_attn_decode_packed', n, _ptrs(qs), _longs([int(q.stride(-2)) for q in qs]), _ptrs([c.kc for c in pk]),
             _ptrs([c.vt for c in pk]), _ptrs([c.mask for c in pk]), _ints([c.Sp for c in pk]),
             _ptrs([a_all[i] for i in range(n)]), _longs([E] * n), M, mods[0].num_heads, M // pk[0].Bc)
        return a_all
    ks, vs, k_ss, k_sb, v_ss, v_sb, masks, S, bk, bv = [], [], [], [], [], [], [], [], [], []
    k_sh, v_sh = [], []
    beams = 1
    for i, (nm, m) in enumerate(zip(names, mods)):
        k, v = kv_layer[nm]
        if k.shape[0] == 0:                                     # empty context: bias and zero rows only
            ks.append(qs[i]); vs.append(qs[i]); masks.append(None)
            k_ss.append(0); k_sb.append(0); v_ss.append(0); v_sb.append(0); S.append(0)
            k_sh.append(64); v_sh.append(64)
        else:
            assert k.stride(-1) == 1 and v.stride(-1) == 1 and M % k.shape[1] == 0
            beams = M // k.shape[1]
            ks.append(k); vs.append(v)
            k_ss.append(k.stride(0)); k_sb.append(k.stride(1)); v_ss.append(v.stride(0)); v_sb.append(v.stride(1))
            # [S, B, E] (heads side by side in a row) or a head-major cache seen as [S, B, H, 64]
            k_sh.append(k.stride(2) if k.dim() == 4 else 64); v_sh.append(v.stride(2) if v.dim() == 4 else 64)
            S.append(k.shape[0])
            mk = contexts.get(nm + '_mask')
            if mk is not None and mk.dtype != torch.uint8:
                mk = mk.to(torch.uint8)
            masks.append(mk.contiguous() if mk is not None else None)
        bk.append(ops._bias_row(m.bias_k, torch.bfloat16))
        bv.append(ops._bias_row(m.bias_v, torch.bfloat16))
    q_sb = [int(q.stride(-2)) for q in qs]
    call('tell_attn_decode', n, _ptrs(qs), _longs(q_sb), _ptrs(ks), _longs(k_ss), _longs(k_sb), _longs(k_sh), _ptrs(vs),
         _longs(v_ss), _longs(v_sb), _longs(v_sh), _ptrs(masks), _ptrs(bk), _ptrs(bv), 1, _ints(S),
         _ptrs([a_all[i] for i in range(n)]), _longs([E] * n), M, mods[0].num_heads, beams)
    return a_all


def decoder_step(dec, X, contexts, state, kv_cache):
    """X [1, M, E] bf16 (embedded tokens of this step) -> [1, M, E] bf16 after all layers; the DynamicConv input
    buffers in `state` are shifted in place."""
    M, E = X.shape[1], X.shape[2]
    dev = X.device
    x_bf = X.reshape(M, E)
    raw_in, st_in, ln_in = None, None, None           # from layer 1 on: fp32 pre-norm rows of the previous layer
    raw_in_bf = None                                  # ... and their bf16 copy (FOLD: the operand of the next linear1)
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    for li, layer in enumerate(dec.layers):
        conv = layer.conv
        C, H, K = layer.conv_dim, conv.num_heads, conv.kernel_size
        names = layer.context_names
        n = len(names)
        mods = [layer.context_attns[nm] for nm in names]
        fold = FOLD and n in (1, 2, 4)
        last = li + 1 == len(dec.layers)
        # ---- conv block (:256-266): linear1 + GLU | tap projection, softmax, K-tap sum, buffer shift | linear2 + res
        w1, _ = ops.wn_weight(layer.linear1.weight_g, layer.linear1.weight_v)
        g = torch.empty(M, C, **bf)
        if raw_in is None:
            _skinny([x_bf], E, [w1], [layer.linear1.bias], [g], C, M, C, E, act=2)
        elif raw_in_bf is not None:
            st_in = torch.empty(M, 2, **f32)
            w1f, s1, c1 = _folded(layer.linear1.weight_v, w1, [ln_in], E)
            _skinny([raw_in_bf], E, [w1f], [layer.linear1.bias], [g], C, M, C, E, pro=3, gammas=[s1], betas=[c1],
                    eps=ln_in.eps, stats_out=st_in, act=2)
        else:
            st_in = torch.empty(M, 2, **f32)
            _skinny([raw_in], E, [w1], [layer.linear1.bias], [g], C, M, C, E, pro=1, gammas=[ln_in.weight],
                    betas=[ln_in.bias], eps=ln_in.eps, stats_out=st_in, act=2)
        c = torch.empty(M, C, **bf)
        hist = state[conv._state_key]
        assert hist.is_contiguous() and hist.shape[1] == M and hist.shape[0] == K
        call('tell_dynconv_step', g, hist, ops.weight(conv.weight_linear.weight), c, M, C, H, K, int(state['_t_cur']),
             state.get('_back'))
        w2, _ = ops.wn_weight(layer.linear2.weight_g, layer.linear2.weight_v)
        raw3 = torch.empty(M, E, **f32)
        raw3_bf = torch.empty(M, E, **bf) if fold else None
        if raw_in is None:
            _skinny([c], C, [w2], [layer.linear2.bias], [raw3], E, M, E, C, res=x_bf, ld_res=E, out2=raw3_bf, out_f32=True)
        else:
            _skinny([c], C, [w2], [layer.linear2.bias], [raw3], E, M, E, C, res_raw=raw_in, res_stats=st_in,
                    res_ln=ln_in, out2=raw3_bf, out_f32=True)
        # ---- context block (:271-355): n query projections of LN(raw3) | n attentions | n output projections + LN(raw3)
        ln3 = layer.conv_layer_norm
        st3 = torch.empty(M, 2, **f32)
        q_all = torch.empty(n, M, E, **bf)
        wq, bq = [], []
        for m in mods:
            wp, rows = m._wrows(0)
            wq.append(ops.weight(wp, rows))
            bq.append(m.in_proj_bias.detach()[0:E])
        if fold:
            fq = [_folded(m._wrows(0)[0], w_, [ln3], E) for m, w_ in zip(mods, wq)]
            _skinny([raw3_bf] * n, E, [f[0] for f in fq], bq, [q_all[i] for i in range(n)], E, M, E, E, pro=3,
                    gammas=[f[1] for f in fq], betas=[f[2] for f in fq], eps=ln3.eps, stats_out=st3, scale=mods[0].scaling)
        else:
            _skinny([raw3] * n, E, wq, bq, [q_all[i] for i in range(n)], E, M, E, E, pro=1, gammas=[ln3.weight],
                    betas=[ln3.bias], eps=ln3.eps, stats_out=st3, scale=mods[0].scaling)
        a_all = attn_decode_all(mods, names, [q_all[i] for i in range(n)], kv_cache[li], contexts, M, E)
        raw6 = torch.empty(M, n * E, **f32)
        raw6_bf = torch.empty(M, n * E, **bf) if fold else None     # (problem i's copy lands at columns i E ..: out2_prob)
        _skinny([a_all[i] for i in range(n)], E, [ops.weight(m.out_proj.weight) for m in mods],
                [m.out_proj.bias for m in mods], [raw6[:, i * E:(i + 1) * E] for i in range(n)], n * E, M, E, E,
                res_raw=raw3, res_stats=st3, res_ln=ln3, out2=raw6_bf, out_f32=True)
        lns = [layer.context_attn_lns[nm] for nm in names]
        wc, _ = ops.wn_weight(layer.context_fc.weight_g, layer.context_fc.weight_v)
        x2 = torch.empty(M, E, **bf)
        if fold:
            wcf, sc, cc = _folded(layer.context_fc.weight_v, wc, lns, E)
            _skinny([raw6_bf], n * E, [wcf], [layer.context_fc.bias], [x2], E, M, E, n * E, pro=4, gammas=[sc], betas=[cc],
                    seg=E, eps=lns[0].eps)
        else:
            _skinny([raw6], n * E, [wc], [layer.context_fc.bias], [x2], E, M, E, n * E, pro=2,
                    gammas=[ln.weight for ln in lns], betas=[ln.bias for ln in lns], seg=E, eps=lns[0].eps)
        # ---- FFN (:357-364)
        F = layer.fc1.out_features
        wf1, _ = ops.wn_weight(layer.fc1.weight_g, layer.fc1.weight_v)
        wf2, _ = ops.wn_weight(layer.fc2.weight_g, layer.fc2.weight_v)
        hmid = torch.empty(M, F, **bf)
        _skinny([x2], E, [wf1], [layer.fc1.bias], [hmid], F, M, F, E, act=1)
        raw9 = torch.empty(M, E, **f32)
        nxt_fold = FOLD and not last and len(dec.layers[li + 1].context_names) in (1, 2, 4)
        raw9_bf = torch.empty(M, E, **bf) if nxt_fold else None
        _skinny([hmid], F, [wf2], [layer.fc2.bias], [raw9], E, M, E, F, res=x2, ld_res=E, out2=raw9_bf, out_f32=True)
        raw_in, ln_in, raw_in_bf = raw9, layer.final_layer_norm, raw9_bf
    y = torch.empty(M, E, **bf)
    call('tell_layernorm_rows', raw_in, E, ln_in.weight, ln_in.bias, ln_in.eps, y, E, None, M, E)
    return y.view(1, M, E)
