#!/usr/bin/env python
"""Generate the committed golden fixtures by RUNNING THE REAL REFERENCE.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden.py
Writes small `.npz` files next to this script.  Each file holds
  sd/<state_dict key>  - the reference module's weights,
  in/<name>            - inputs,
  out/<name>           - what the reference computed.
The fixtures are data (inputs + expected outputs); no reference source travels.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import seeded  # noqa: E402

dfo, dfl, tfo, tfl, _ = ref_import.import_models()
dfp, dni = ref_import.import_decoder_variants()
from tell.modules.attention.multi_head import MultiHeadAttention  # noqa: E402
from tell.modules.convolutions.dynamic import DynamicConv1dTBC  # noqa: E402
from tell.modules.convolutions.lightweight import LightweightConv1dTBC  # noqa: E402
from tell.modules.criteria.adaptive_loss import AdaptiveLoss  # noqa: E402
from tell.modules.linear import GehringLinear  # noqa: E402
from tell.modules.softmax import AdaptiveSoftmax  # noqa: E402
from tell.modules.token_embedders.adaptive import AdaptiveEmbedding  # noqa: E402
from tell.modules.token_embedders.positional import (SinusoidalPositionalEmbedding,  # noqa: E402
                                                     make_positions)
from tell.modules.token_embedders.sum_text_field_embedder import SumTextFieldEmbedder  # noqa: E402


_SEEDED = {}


def seed_big(name, tensors, positive=()):
    """Overwrite every large tensor of `tensors` (dict key -> tensor, e.g. named
    parameters or inputs) with seeded values BEFORE the reference runs; `save`
    then stores only the seed record for them."""
    seen = set()
    for grp_key, t in tensors.items():
        if torch.is_tensor(t) and t.is_floating_point() and t.numel() > seeded.THRESH:
            if t.data_ptr() in seen:      # tied weight: one storage, several state_dict keys
                continue
            seen.add(t.data_ptr())
            full = name + '.npz:' + grp_key
            with torch.no_grad():
                _SEEDED[full] = seeded.fill_(t, full, positive=grp_key in positive)


def save(name, sd=None, **groups):
    flat = {}
    groups = dict(groups)
    if sd is not None:
        groups['sd'] = sd
    for g, d in groups.items():
        first = {}
        for k, v in d.items():
            full = '%s.npz:%s/%s' % (name, g, k)
            if torch.is_tensor(v) and v.numel() > 1:
                if v.data_ptr() in first:             # tied duplicate -> store an alias only
                    flat['alias_%s/%s' % (g, k)] = np.array(first[v.data_ptr()])
                    continue
                first[v.data_ptr()] = k
            if full in _SEEDED:
                chk = seeded.regen(full, _SEEDED[full])
                assert torch.equal(chk, v.detach().float()), full
                flat['seeded_%s/%s' % (g, k)] = _SEEDED[full]
                continue
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            v = np.asarray(v)
            if g == 'out' and v.size > seeded.SUB_N:      # large expected outputs: strided subsample
                flat['sub_%s/%s' % (g, k)] = seeded.subsample(v)
                continue
            flat['%s/%s' % (g, k)] = v
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **flat)
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def gehring():
    torch.manual_seed(1)
    m = GehringLinear(24, 40, dropout=0.1)
    m.weight_g.data.mul_(torch.rand(40, 1) + 0.5)
    m.bias.data.normal_()
    x = torch.randn(5, 3, 24, requires_grad=True)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    save('gehring_linear', m.state_dict(), **{'in': {'x': x, 'gy': gy},
         'out': {'y': y, 'gx': x.grad, 'g_weight_g': m.weight_g.grad, 'g_weight_v': m.weight_v.grad,
                 'g_bias': m.bias.grad}})


def dynconv():
    for K, T in ((3, 6), (7, 4), (31, 12), (15, 40)):
        torch.manual_seed(10 + K)
        m = DynamicConv1dTBC(64, K, padding_l=K - 1, num_heads=4, weight_softmax=True,
                             weight_dropout=0.1).eval()
        x = torch.randn(T, 2, 64, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        # incremental: feed one step at a time (the generation path, dynamic.py:94-99,115-116)
        st = {}
        inc = torch.cat([m(x[t:t + 1].detach(), incremental_state=st) for t in range(T)], dim=0)
        # incremental with a 2-row first chunk
        st2 = {}
        inc2 = torch.cat([m(x[0:2].detach(), incremental_state=st2)] +
                         [m(x[t:t + 1].detach(), incremental_state=st2) for t in range(2, T)], dim=0)
        save('dynconv_K%d_T%d' % (K, T), m.state_dict(),
             **{'in': {'x': x, 'gy': gy},
                'out': {'y': y, 'gx': x.grad, 'g_weight': m.weight_linear.weight.grad,
                        'y_incremental': inc, 'y_incremental2': inc2}})


def lightconv():
    """LightweightConv1dTBC (static taps per head, lightweight.py:83-240) as `decoder_conv_type: lightweight` builds it."""
    for K, T in ((3, 6), (31, 12)):
        torch.manual_seed(20 + K)
        m = LightweightConv1dTBC(64, K, padding_l=K - 1, num_heads=4, weight_softmax=True,
                                 weight_dropout=0.1).eval()
        x = torch.randn(T, 2, 64, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        st = {}
        inc = torch.cat([m(x[t:t + 1].detach(), incremental_state=st) for t in range(T)], dim=0)
        save('lightconv_K%d_T%d' % (K, T), m.state_dict(),
             **{'in': {'x': x, 'gy': gy},
                'out': {'y': y, 'gx': x.grad, 'g_weight': m.weight.grad, 'y_incremental': inc}})


def mha():
    cases = [('sep', 48, 6, True), ('same', 64, 9, True), ('nomask', 48, 5, False), ('empty', 0, 1, True)]
    for tag, kdim, S, use_mask in cases:
        torch.manual_seed(20 + S)
        m = MultiHeadAttention(64, 4, kdim=kdim if kdim else 16, vdim=kdim if kdim else 16,
                               dropout=0.1).eval()
        if kdim == 0:
            # the reader's empty face list: feature dim 0 -> k = v = bias rows only
            # (multi_head.py:349-374); projection weights exist but are unused
            key = torch.zeros(S, 2, 0)
        else:
            key = torch.randn(S, 2, kdim)
        m.in_proj_bias.data.normal_()
        m.out_proj.bias.data.normal_()
        q = torch.randn(3, 2, 64, requires_grad=True)
        mask = None
        if use_mask:
            mask = torch.zeros(2, S, dtype=torch.bool)
            if S > 2:
                mask[0, -2:] = True
                mask[1, 1] = True
        y, w = m(q, key, key, key_padding_mask=mask, incremental_state=None, static_kv=True,
                 need_weights=True)
        gy = torch.randn_like(y)
        y.backward(gy)
        grads = {'g_' + k: v.grad for k, v in m.named_parameters() if v.grad is not None}
        ins = {'q': q, 'key': key, 'gy': gy, 'kdim': kdim}
        if mask is not None:
            ins['mask'] = mask
        save('mha_' + tag, m.state_dict(), **{'in': ins, 'out': dict(y=y, w=w, gq=q.grad, **grads)})


def _ref_embedder(V, E, cutoff, init_size=16):
    return SumTextFieldEmbedder(
        {'adaptive': AdaptiveEmbedding(None, 'bpe', 0, E, 1, E, list(cutoff), vocab_size=V,
                                       scale_embeds=True),
         'position': SinusoidalPositionalEmbedding(None, E, 1, False, init_size=init_size)},
        embedder_to_indexer_map={'adaptive': ['roberta'], 'position': ['roberta']},
        allow_unmatched_keys=True)


def embed_and_positions():
    torch.manual_seed(30)
    V, E, cutoff = 600, 32, (100, 300)
    emb = _ref_embedder(V, E, cutoff, init_size=8)
    ids = torch.randint(2, V, (3, 7))
    ids[:, 0] = 0
    ids[0, 1:4] = torch.tensor([99, 100, 300])       # band edges
    ids[1, 5:] = 1                                   # right padding
    ids[2, 2] = 299
    y = emb({'roberta': ids})
    gy = torch.randn_like(y)
    y.backward(gy)
    grads = {'g_' + k: v.grad for k, v in emb.named_parameters()}
    # incremental position offsets (positional.py:170-173).  NOTE: the reference's table
    # growth (:180-187) allocates one row too few and raises IndexError when it triggers
    # (never with the configs' init_size 512), so it is not exercised here.
    st = {}
    inc = [emb({'roberta': ids[:, t:t + 1]}, incremental_state=st) for t in range(7)]
    # reference's own known-answer vectors (tests/test_positional.py:8-44) are re-run here
    pos_l = make_positions(torch.LongTensor([[9, 9, 9, 9, 9], [1, 9, 9, 9, 9], [1, 1, 1, 9, 9]]), 1, True)
    pos_r = make_positions(torch.LongTensor([[9, 9, 9, 9, 9], [9, 9, 9, 9, 1], [9, 9, 1, 1, 1]]), 1, False)
    save('embedder', {k: v for k, v in emb.state_dict().items() if 'position' not in k},
         **{'in': {'ids': ids, 'gy': gy, 'cutoff': np.array(cutoff), 'V': V},
            'out': dict(y=y, y_incremental=torch.cat(inc, dim=1), pos_left=pos_l,
                        pos_right=pos_r, **grads)})


def adaptive_softmax():
    torch.manual_seed(40)
    V, E, cutoff = 600, 32, (100, 300)
    emb = _ref_embedder(V, E, cutoff)
    asm = AdaptiveSoftmax(V, E, list(cutoff), dropout=0, adaptive_inputs=emb.token_embedder_adaptive,
                          factor=1, tie_proj=False)
    crit = AdaptiveLoss(padding_idx=1)
    x = torch.randn(4, 6, E, requires_grad=True)
    tgt = torch.randint(2, V, (4, 6))
    tgt[0, :4] = torch.tensor([99, 100, 101, 301])    # 101/301: tail-local index 1 -> ignored in tail
    tgt[1, 4:] = 1                                    # pads
    tgt[2, :] = torch.randint(2, 100, (6,))           # row with head-only targets
    loss, n = crit(asm, (x, None), tgt)
    loss.backward()
    grads = {'g_' + k: v.grad for k, v in asm.named_parameters() if v.grad is not None}
    lp = asm.get_log_prob(x.detach(), None)
    # a batch with NO target in band 2 (softmax.py:160-165 `None` branch)
    tgt2 = torch.randint(2, 300, (4, 6))
    loss2, n2 = crit(asm, (x.detach(), None), tgt2)
    save('adaptive_softmax', asm.state_dict(),
         **{'in': {'x': x, 'target': tgt, 'target2': tgt2, 'cutoff': np.array(cutoff), 'V': V},
            'out': dict(loss=loss, sample_size=n, loss2=loss2, sample_size2=n2, log_probs=lp,
                        gx=x.grad, **grads)})


ART_DIM = {'flattened': 64, 'faces_objects': 1024, 'faces_parallel': 1024, 'flattened_no_image': 64,
           'flattened_lightweight': 64, 'flattened_prenorm': 64}   # flattened: kdim == embed_dim -> in_proj_weight path


def lstm_decoder():
    """LSTMDecoder (`lstm_decoder_flattened`, decoder_flattened_lstm.py:68-208): the decoder of the GloVe/LSTM baseline."""
    import importlib
    dl = importlib.import_module('tell.models.decoder_flattened_lstm')
    torch.manual_seed(60)
    emb = _ref_embedder(600, 64, (100, 300), init_size=512)
    dec = dl.LSTMDecoder(None, emb, num_layers=3, hidden_size=48, dropout=0.1, share_decoder_input_output_embed=True,
                         vocab_size=600, adaptive_softmax_cutoff=[100, 300], tie_adaptive_weights=True,
                         adaptive_softmax_dropout=0, tie_adaptive_proj=False, adaptive_softmax_factor=1,
                         article_embed_size=300, image_embed_size=2048).eval()
    for p in dec.parameters():
        if p.dim() == 1 or p.shape[0] == 1:          # biases, learned initial states
            p.data.add_(0.1 * torch.randn_like(p))
    seed_big('decoder_lstm', {'sd/' + k: v for k, v in dec.state_dict().items() if 'token_embedder_position' not in k})
    B, T, S = 2, 7, 11
    g = torch.Generator().manual_seed(61)
    ctx = {'image': torch.randn(5, B, 2048, generator=g), 'image_mask': torch.zeros(B, 5, dtype=torch.bool),
           'article': torch.randn(S, B, 300, generator=g), 'article_mask': torch.zeros(B, S, dtype=torch.bool)}
    ctx['article_mask'][0, S - 3:] = True
    seed_big('decoder_lstm', {'in/' + k: v for k, v in ctx.items()})
    ids = torch.randint(2, 600, (B, T))
    ids[:, 0] = 0
    ids[1, 5:] = 1
    tgt = torch.randint(2, 600, (B, T))
    tgt[1, 4:] = 1
    tgt[0, :3] = torch.tensor([101, 301, 99])
    out = dec({'roberta': ids}, ctx)
    loss, n = AdaptiveLoss(padding_idx=1)(dec.adaptive_softmax, out, tgt)
    (loss / n).backward()
    pd = dict(dec.named_parameters())
    names = ['layers.0.weight_ih', 'layers.0.bias_hh', 'layers.2.weight_hh', 'h.0', 'c.1',
             'image_attention.input_proj.weight_v', 'image_attention.input_proj.bias',
             'article_attention.output_proj.weight_g', 'article_attention.output_proj.bias', 'attn_proj.weight_v',
             'project_out_dim.weight_v', 'embedder.token_embedder_adaptive.embeddings.0.0.weight',
             'adaptive_softmax.tail.1.0.weight']
    grads = {'g_' + k: pd[k].grad for k in names}
    sd = {k: v for k, v in dec.state_dict().items() if 'token_embedder_position' not in k}
    save('decoder_lstm', sd, **{'in': dict(ids=ids, target=tgt, **ctx),
                                'out': dict(x=out[0], loss=loss, sample_size=n, **grads)})


def baseline_model():
    """BaselineGloveModel + LSTMDecoder (`baseline_glove`, expt/*/1_lstm_glove): loss and greedy generation.  spaCy is
    absent; the stand-in below supplies a token list with `.has_vector` / `.vector` per article (deterministic vectors
    per token id), and the NaN-padded tensor the reference builds from it (:207-220) is stored as the model INPUT."""
    import importlib
    import types
    import numpy as np

    class _Tok:
        def __init__(self, w):
            i = int(w[1:])
            self.has_vector = i % 5 != 0
            self.vector = np.random.RandomState(1000 + i).randn(300).astype('float32')

    class _NLP:
        def pipe(self, texts):
            for t in texts:
                yield [_Tok(w) for w in t.split()]
    sp = types.ModuleType('spacy')
    sp.load = lambda *a, **k: _NLP()
    sys.modules['spacy'] = sp
    bg = importlib.import_module('tell.models.baseline_glove')
    dl = importlib.import_module('tell.models.decoder_flattened_lstm')
    torch.manual_seed(70)
    emb = _ref_embedder(600, 64, (100, 300), init_size=512)
    dec = dl.LSTMDecoder(None, emb, num_layers=2, hidden_size=48, dropout=0.1, share_decoder_input_output_embed=True,
                         vocab_size=600, adaptive_softmax_cutoff=[100, 300], tie_adaptive_weights=True,
                         adaptive_softmax_dropout=0, tie_adaptive_proj=False, adaptive_softmax_factor=1,
                         article_embed_size=300, image_embed_size=2048)
    model = bg.BaselineGloveModel(None, dec, AdaptiveLoss(padding_idx=1)).eval()
    for p in dec.parameters():
        if p.dim() == 1 or p.shape[0] == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    w0 = dec.embedder.token_embedder_adaptive.embeddings[0][0].weight
    w0.data[2] *= 1.3                       # <eos> row: greedy decoding ends at different steps per row
    seed_big('model_baseline_glove', {'sd/' + k: v for k, v in model.state_dict().items()
                                      if not k.startswith(('resnet.', 'roberta.')) and 'token_embedder_position' not in k
                                      and 'embeddings.0.0' not in k and 'head.word_proj' not in k})
    B = 3
    g = torch.Generator().manual_seed(71)
    image = torch.randn(B, 3, 224, 224, generator=g)
    seed_big('model_baseline_glove', {'in/image': image})
    lens = [9, 14, 6]
    texts = [' '.join('W%d' % int(t) for t in torch.randint(1, 400, (n,), generator=g)) for n in lens]
    cap = torch.randint(4, 600, (B, 8), generator=g)
    cap[:, 0] = 0
    cap[1, 6:] = 1
    cap[1, 5] = 2
    # the tensor :207-220 builds (lower-cased text -> our stand-in parses 'w<i>')
    vs = [[_Tok(w).vector for w in t.lower().split() if _Tok(w).has_vector] for t in texts]
    L = max(len(v) for v in vs)
    cv = torch.full((B, L, 300), float('nan'))
    for i, v in enumerate(vs):
        cv[i, :len(v)] = torch.from_numpy(np.array(v))
    meta = [{'context': t} for t in texts]
    out = model(image.clone(), {'roberta': cap.clone()}, meta)
    cid, tid, ctx = model._forward(texts, image.clone(), {'roberta': cap.clone()})
    assert torch.equal(torch.nan_to_num(cv), ctx['article'].transpose(0, 1)), 'stand-in tensor != what the reference built'
    lp, gen_ids = model._generate(cid, ctx)
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith(('resnet.', 'roberta.'))
          and 'token_embedder_position' not in k}
    save('model_baseline_glove', sd, **{'in': dict(image=image, caption=cap, context_vectors=cv),
                                        'out': dict(loss=out['loss'], sample_size=out['sample_size'], gen_ids=gen_ids,
                                                    gen_log_probs=lp)})
    print('   gen lengths:', [(r != 1).sum().item() for r in gen_ids], 'gen shape', tuple(gen_ids.shape))


def transformer_glove_model():
    """TransformerGloveModel (`transformer_glove`, expt/*/2_transformer_glove): the flattened decoder over GloVe vectors."""
    import importlib
    import types
    import numpy as np

    class _Tok:
        def __init__(self, w):
            i = int(w[1:])
            self.has_vector = i % 5 != 0
            self.vector = np.random.RandomState(1000 + i).randn(300).astype('float32')

    class _NLP:
        def pipe(self, texts):
            for t in texts:
                yield [_Tok(w) for w in t.split()]
    sp = types.ModuleType('spacy')
    sp.load = lambda *a, **k: _NLP()
    sys.modules['spacy'] = sp
    tg = importlib.import_module('tell.models.transformer_glove')
    torch.manual_seed(80)
    emb = _ref_embedder(600, 64, (100, 300), init_size=512)
    dec = dfl.DynamicConvDecoder(None, emb, article_embed_size=300, **DEC_KW)
    model = tg.TransformerGloveModel(None, dec, AdaptiveLoss(padding_idx=1), vocab_size=600).eval()
    for p in dec.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    w0 = dec.embedder.token_embedder_adaptive.embeddings[0][0].weight
    w0.data[2] *= 1.6
    seed_big('model_transformer_glove', {'sd/' + k: v for k, v in model.state_dict().items()
                                         if not k.startswith(('resnet.', 'roberta.')) and 'token_embedder_position' not in k
                                         and 'embeddings.0.0' not in k and 'head.word_proj' not in k})
    B = 3
    g = torch.Generator().manual_seed(81)
    image = torch.randn(B, 3, 224, 224, generator=g)
    seed_big('model_transformer_glove', {'in/image': image})
    lens = [9, 14, 6]
    texts = [' '.join('W%d' % int(t) for t in torch.randint(1, 400, (n,), generator=g)) for n in lens]
    cap = torch.randint(4, 600, (B, 8), generator=g)
    cap[:, 0] = 0
    cap[1, 6:] = 1
    cap[1, 5] = 2
    vs = [[_Tok(w).vector for w in t.lower().split() if _Tok(w).has_vector] for t in texts]
    L = max(len(v) for v in vs)
    cv = torch.full((B, L, 300), float('nan'))
    for i, v in enumerate(vs):
        cv[i, :len(v)] = torch.from_numpy(np.array(v))
    out = model(image.clone(), {'roberta': cap.clone()}, [{'context': t} for t in texts])
    cid, tid, ctx = model._forward(texts, image.clone(), {'roberta': cap.clone()})
    assert torch.equal(torch.nan_to_num(cv), ctx['article'].transpose(0, 1))
    lp, gen_ids = model._generate(cid, ctx)[:2]
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith(('resnet.', 'roberta.'))
          and 'token_embedder_position' not in k}
    save('model_transformer_glove', sd, **{'in': dict(image=image, caption=cap, context_vectors=cv),
                                           'out': dict(loss=out['loss'], sample_size=out['sample_size'], gen_ids=gen_ids,
                                                       gen_log_probs=lp)})
    print('   gen lengths:', [(r != 1).sum().item() for r in gen_ids], 'gen shape', tuple(gen_ids.shape))


def _mk_contexts(B, S, kind, seed):
    g = torch.Generator().manual_seed(seed)
    ctx = {'image': torch.randn(5, B, 2048, generator=g), 'image_mask': torch.zeros(B, 5, dtype=torch.bool),
           'article': torch.randn(S, B, ART_DIM[kind], generator=g),
           'article_mask': torch.zeros(B, S, dtype=torch.bool)}
    ctx['article_mask'][0, S - 3:] = True
    if kind == 'flattened_no_image':
        del ctx['image'], ctx['image_mask']
    if kind in ('faces_objects', 'faces_parallel'):
        ctx['faces'] = torch.randn(3, B, 512, generator=g)
        ctx['faces_mask'] = torch.zeros(B, 3, dtype=torch.bool)
        ctx['faces_mask'][1, 1:] = True
    if kind == 'faces_objects':
        ctx['obj'] = torch.randn(6, B, 2048, generator=g).abs()
        ctx['obj_mask'] = torch.zeros(B, 6, dtype=torch.bool)
        ctx['obj_mask'][0, 4:] = True
    return ctx


DEC_KW = dict(max_target_positions=512, dropout=0.1, share_decoder_input_output_embed=True,
              decoder_output_dim=64, decoder_conv_dim=64, decoder_glu=True, decoder_conv_type='dynamic',
              weight_softmax=True, decoder_attention_heads=4, weight_dropout=0.1, relu_dropout=0.0,
              input_dropout=0.1, decoder_normalize_before=False, attention_dropout=0.1,
              decoder_ffn_embed_dim=128, decoder_kernel_size_list=[3, 7, 15, 31],
              adaptive_softmax_cutoff=[100, 300], adaptive_softmax_factor=1, tie_adaptive_weights=True,
              adaptive_softmax_dropout=0, tie_adaptive_proj=False, decoder_layers=4, final_norm=False,
              padding_idx=0, namespace='bpe', vocab_size=600)


def _ref_decoder(kind):
    emb = _ref_embedder(600, 64, (100, 300), init_size=512)
    if kind == 'flattened_prenorm':           # pre-LN blocks + final LayerNorm, no GLU (the remaining layer switches)
        return dfl.DynamicConvDecoder(None, emb, article_embed_size=ART_DIM['flattened'],
                                      **dict(DEC_KW, decoder_normalize_before=True, final_norm=True, decoder_glu=False))
    if kind == 'flattened_lightweight':       # the 2-context decoder with `decoder_conv_type: lightweight`
        return dfl.DynamicConvDecoder(None, emb, article_embed_size=ART_DIM['flattened'],
                                      **dict(DEC_KW, decoder_conv_type='lightweight'))
    if kind == 'faces_objects':
        return dfo.DynamicConvFacesObjectsDecoder(None, emb, **DEC_KW)
    if kind == 'faces_parallel':
        return dfp.DynamicConvFacesParallelDecoder(None, emb, **DEC_KW)
    if kind == 'flattened_no_image':
        return dni.DynamicConvDecoderNoImage(None, emb, article_embed_size=ART_DIM[kind], **DEC_KW)
    return dfl.DynamicConvDecoder(None, emb, article_embed_size=ART_DIM[kind], **DEC_KW)


def decoders(kinds=('flattened', 'faces_objects', 'faces_parallel', 'flattened_no_image')):
    for kind in kinds:
        torch.manual_seed(50)
        dec = _ref_decoder(kind).eval()
        for p in dec.parameters():          # make biases / LN params non-trivial
            if p.dim() == 1:
                p.data.add_(0.1 * torch.randn_like(p))
        seed_big('decoder_' + kind, {'sd/' + k: v for k, v in dec.state_dict().items()
                                     if 'token_embedder_position' not in k})
        B, T, S = 2, 9, 11
        ctx = _mk_contexts(B, S, kind, 51)
        seed_big('decoder_' + kind, {'in/' + k: v for k, v in ctx.items()}, positive=('in/obj',))
        ids = torch.randint(2, 600, (B, T))
        ids[:, 0] = 0
        ids[1, 6:] = 1
        tgt = torch.randint(2, 600, (B, T))
        tgt[1, 5:] = 1
        tgt[0, :3] = torch.tensor([101, 301, 99])
        crit = AdaptiveLoss(padding_idx=1)
        out = dec({'roberta': ids}, ctx)
        loss, n = crit(dec.adaptive_softmax, out, tgt)
        (loss / n).backward()
        names = ['layers.0.linear1.weight_v', 'layers.0.linear1.weight_g',
                 'layers.3.conv.weight' if kind == 'flattened_lightweight' else 'layers.3.conv.weight_linear.weight',
                 'layers.1.context_attns.article.' + ('in_proj_weight' if kind.startswith('flattened') else 'v_proj_weight'),
                 'layers.1.context_attns.' + ('article.in_proj_bias' if kind == 'flattened_no_image' else 'image.k_proj_weight'),
                 'layers.2.context_attns.' + ('article' if kind == 'flattened_no_image' else 'image') + '.bias_k',
                 'layers.2.context_attn_lns.article.weight',
                 'layers.3.fc2.bias', 'layers.0.context_fc.weight_v',
                 'embedder.token_embedder_adaptive.embeddings.0.0.weight',
                 'embedder.token_embedder_adaptive.embeddings.2.1.weight',
                 'adaptive_softmax.tail.1.0.weight', 'adaptive_softmax.head.class_proj.weight']
        pd = dict(dec.named_parameters())
        grads = {'g_' + k: pd[k].grad for k in names}
        # eval-mode attention weights of layer 0 (need_weights branch, multi_head.py:478-482)
        attn0 = {'attn0_' + k: v for k, v in out[1]['attn'][0].items()} if kind == 'faces_objects' else {}
        # incremental decode == full decode (teacher forced)
        st = {}
        inc = torch.cat([dec({'roberta': ids[:, t:t + 1]}, ctx, incremental_state=st)[0]
                         for t in range(T)], dim=1)
        sd = {k: v for k, v in dec.state_dict().items() if 'token_embedder_position' not in k}
        ins = dict(ids=ids, target=tgt, **ctx)
        save('decoder_' + kind, sd, **{'in': ins, 'out': dict(x=out[0], loss=loss, sample_size=n,
                                                             x_incremental=inc, **grads, **attn0)})


def models():
    for kind in ('flattened', 'faces_objects'):
        torch.manual_seed(60)
        dec = _ref_decoder(kind)
        crit = AdaptiveLoss(padding_idx=1)
        if kind == 'faces_objects':
            model = tfo.TransformerFacesObjectModel(None, dec, crit, weigh_bert=True, vocab_size=600)
        else:
            model = tfl.TransformerFlattenedModel(None, dec, crit, weigh_bert=True, vocab_size=600)
            model.roberta = ref_import.StandInEncoders.Roberta(dim=ART_DIM[kind])
        model.eval()
        seed_big('model_' + kind, {'sd/' + k: v for k, v in model.state_dict().items()
                                   if not k.startswith(('resnet.', 'roberta.'))
                                   and 'token_embedder_position' not in k and 'embeddings.0.0' not in k and 'head.word_proj' not in k})
        # bias the tied <eos>=2 output row so that greedy decoding terminates at
        # different steps for different rows (exercises active-row compaction)
        w0 = dec.embedder.token_embedder_adaptive.embeddings[0][0].weight
        w0.data[2] *= (1.4 if kind == 'faces_objects' else 1.05)
        B = 4
        g = torch.Generator().manual_seed(61)
        image = torch.randn(B, 3, 224, 224, generator=g)
        seed_big('model_' + kind, {'in/image': image})
        art = torch.randint(4, 600, (B, 20), generator=g)
        art[:, 0] = 0
        art[2, 15:] = 1
        art[2, 14] = 2
        cap = torch.randint(4, 600, (B, 8), generator=g)
        cap[:, 0] = 0
        cap[1, 6:] = 1
        cap[1, 5] = 2
        batch = dict(context={'roberta': art.clone()}, image=image, caption={'roberta': cap.clone()})
        if kind == 'faces_objects':
            faces = torch.randn(B, 4, 512, generator=g)
            objs = torch.randn(B, 7, 2048, generator=g).abs()
            seed_big('model_' + kind, {'in/obj_embeds': objs}, positive=('in/obj_embeds',))
            raw = {'face_embeds': faces.clone(), 'obj_embeds': objs.clone()}
            n_faces, n_objs = [2, 4, 4, 0], [7, 3, 7, 7]      # rows beyond the count are NaN padding
            for b in range(B):
                faces[b, n_faces[b]:] = float('nan')
                objs[b, n_objs[b]:] = float('nan')
            batch.update(face_embeds=faces.clone(), obj_embeds=objs.clone())
        out = model(metadata=[{}] * B, **{k: (v.clone() if torch.is_tensor(v) else dict(v)) for k, v in batch.items()})
        gen_in = {k: (v.clone() if torch.is_tensor(v) else {'roberta': v['roberta'].clone()}) for k, v in batch.items()}
        if kind == 'faces_objects':
            cid, tid, ctx = model._forward(gen_in['context'], gen_in['image'], gen_in['caption'],
                                           gen_in['face_embeds'], gen_in['obj_embeds'])
            lp, gen_ids, _ = model._generate(cid, ctx)
        else:
            cid, tid, ctx = model._forward(gen_in['context'], gen_in['image'], gen_in['caption'])
            lp, gen_ids = model._generate(cid, ctx)[:2]
        sd = {k: v for k, v in model.state_dict().items()
              if not k.startswith('resnet.') and not k.startswith('roberta.')
              and 'token_embedder_position' not in k}
        ins = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        if kind == 'faces_objects':
            ins.update(raw)
            ins['n_faces'], ins['n_objs'] = np.array(n_faces), np.array(n_objs)
        ins['article_ids'] = art
        ins['caption_ids'] = cap
        save('model_' + kind, sd, **{'in': ins, 'out': dict(loss=out['loss'], sample_size=out['sample_size'],
                                                           gen_ids=gen_ids, gen_log_probs=lp)})
        print('   gen lengths:', [(r != 1).sum().item() for r in gen_ids], 'gen shape', tuple(gen_ids.shape))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'transformer_glove':
    transformer_glove_model()
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'lstm':
    lstm_decoder()
    baseline_model()
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'prenorm':
    decoders(('flattened_prenorm',))
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'lightweight':
    lightconv()                                              # only the fixtures added later
    decoders(('flattened_lightweight',))
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'decoder_variants':
    decoders(('faces_parallel', 'flattened_no_image'))       # only the fixtures added later
    sys.exit(0)

if __name__ == '__main__':
    gehring()
    dynconv()
    lightconv()
    mha()
    embed_and_positions()
    adaptive_softmax()
    decoders()
    decoders(('flattened_lightweight', 'flattened_prenorm'))
    lstm_decoder()
    baseline_model()
    transformer_glove_model()
    models()
