"""Pin the oracle (CPU restatement) against fixtures produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import seeded
from oracle import functional as OF
from oracle.build import build_decoder, build_embedder, build_model
from oracle.modules import (AdaptiveLoss, AdaptiveSoftmax, DynamicConv1dTBC, GehringLinear,
                            MultiHeadAttention)

TOL = dict(rtol=1e-4, atol=2e-5)


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    torch.testing.assert_close(torch.as_tensor(a).float(), torch.as_tensor(b).float(), **tol)


def check_outputs(fx, got):
    """Compare dict of results with the fixture's 'out' (full) and 'sub' (subsampled)."""
    for k, v in fx.get('out', {}).items():
        if k in got:
            close(got[k], v)
    for k, v in fx.get('sub', {}).items():
        if k in got:
            close(seeded.subsample(got[k].detach().numpy()), v)


def load_sd(module, sd, strict=True):
    own = module.state_dict()
    sd = {k: v for k, v in sd.items() if k in own}
    missing = [k for k in own if k not in sd and 'token_embedder_position' not in k]
    assert not missing, missing
    module.load_state_dict(sd, strict=False)


def test_gehring_linear(golden):
    fx = golden('gehring_linear')
    m = GehringLinear(24, 40)
    load_sd(m, fx['sd'])
    x = fx['in']['x'].clone().requires_grad_(True)
    y = m(x)
    y.backward(fx['in']['gy'])
    check_outputs(fx, dict(y=y, gx=x.grad, g_weight_g=m.weight_g.grad, g_weight_v=m.weight_v.grad,
                           g_bias=m.bias.grad))


@pytest.mark.parametrize('K,T', [(3, 6), (7, 4), (31, 12), (15, 40)])
def test_dynamic_conv(golden, K, T):
    fx = golden('dynconv_K%d_T%d' % (K, T))
    m = DynamicConv1dTBC(64, K, 4, weight_dropout=0.1).eval()
    load_sd(m, fx['sd'])
    x = fx['in']['x'].clone().requires_grad_(True)
    y = m(x)
    y.backward(fx['in']['gy'])
    st = {}
    inc = torch.cat([m(x[t:t + 1].detach(), incremental_state=st) for t in range(T)], dim=0)
    st2 = {}
    inc2 = torch.cat([m(x[0:2].detach(), incremental_state=st2)] +
                     [m(x[t:t + 1].detach(), incremental_state=st2) for t in range(2, T)], dim=0)
    check_outputs(fx, dict(y=y, gx=x.grad, g_weight=m.weight_linear.weight.grad, y_incremental=inc,
                           y_incremental2=inc2))


@pytest.mark.parametrize('K,T', [(3, 6), (31, 12)])
def test_lightweight_conv(golden, K, T):
    from oracle.modules import LightweightConv1dTBC
    fx = golden('lightconv_K%d_T%d' % (K, T))
    m = LightweightConv1dTBC(64, K, 4, weight_dropout=0.1).eval()
    load_sd(m, fx['sd'])
    x = fx['in']['x'].clone().requires_grad_(True)
    y = m(x)
    y.backward(fx['in']['gy'])
    st = {}
    inc = torch.cat([m(x[t:t + 1].detach(), incremental_state=st) for t in range(T)], dim=0)
    check_outputs(fx, dict(y=y, gx=x.grad, g_weight=m.weight.grad, y_incremental=inc))


@pytest.mark.parametrize('tag', ['sep', 'same', 'nomask', 'empty'])
def test_multi_head_attention(golden, tag):
    fx = golden('mha_' + tag)
    kdim = int(fx['in']['kdim'])
    m = MultiHeadAttention(64, 4, kdim=kdim if kdim else 16, vdim=kdim if kdim else 16,
                           dropout=0.1).eval()
    load_sd(m, fx['sd'])
    q = fx['in']['q'].clone().requires_grad_(True)
    y, w = m(q, fx['in']['key'], fx['in'].get('mask'), need_weights=True)
    y.backward(fx['in']['gy'])
    got = dict(y=y, w=w, gq=q.grad)
    got.update({'g_' + k: v.grad for k, v in m.named_parameters() if v.grad is not None})
    n_grads = sum(1 for k in list(fx['out']) + list(fx.get('sub', {})) if k.startswith('g_'))
    assert n_grads == sum(1 for k in got if k.startswith('g_')) or tag == 'empty'
    check_outputs(fx, got)


def test_make_positions_reference_known_answers(golden):
    """tell/modules/token_embedders/tests/test_positional.py:8-44 known answers."""
    left_in = torch.LongTensor([[9, 9, 9, 9, 9], [1, 9, 9, 9, 9], [1, 1, 1, 9, 9]])
    left_out = torch.LongTensor([[2, 3, 4, 5, 6], [1, 2, 3, 4, 5], [1, 1, 1, 2, 3]])
    right_in = torch.LongTensor([[9, 9, 9, 9, 9], [9, 9, 9, 9, 1], [9, 9, 1, 1, 1]])
    right_out = torch.LongTensor([[2, 3, 4, 5, 6], [2, 3, 4, 5, 1], [2, 3, 1, 1, 1]])
    assert torch.equal(OF.make_positions(left_in, 1, True), left_out)
    assert torch.equal(OF.make_positions(right_in, 1, False), right_out)
    fx = golden('embedder')
    assert torch.equal(fx['out']['pos_left'], left_out)      # what the reference itself returned
    assert torch.equal(fx['out']['pos_right'], right_out)


def test_embedder(golden):
    fx = golden('embedder')
    emb = build_embedder(600, 32, (100, 300), init_size=8)
    load_sd(emb, fx['sd'])
    ids = fx['in']['ids']
    y = emb({'roberta': ids})
    y.backward(fx['in']['gy'])
    got = dict(y=y)
    got.update({'g_' + k: v.grad for k, v in emb.named_parameters()})
    st = {}
    got['y_incremental'] = torch.cat(
        [emb({'roberta': ids[:, t:t + 1]}, incremental_state=st) for t in range(ids.shape[1])], dim=1)
    check_outputs(fx, got)
    # padding_idx rows of every band receive no gradient from the lookup path (adaptive.py:42)
    for i in range(3):
        assert got['g_token_embedder_adaptive.embeddings.%d.0.weight' % i][0].abs().max() == 0


def test_adaptive_softmax_and_loss(golden):
    fx = golden('adaptive_softmax')
    emb = build_embedder(600, 32, (100, 300))
    asm = AdaptiveSoftmax(600, 32, [100, 300], emb.token_embedder_adaptive)
    load_sd(asm, fx['sd'])
    crit = AdaptiveLoss(1)
    x = fx['in']['x'].clone().requires_grad_(True)
    loss, n = crit(asm, (x, None), fx['in']['target'])
    loss.backward()
    assert n == fx['out']['sample_size']
    got = dict(loss=loss.reshape(1), gx=x.grad, log_probs=asm.get_log_prob(x.detach()))
    got.update({'g_' + k: v.grad for k, v in asm.named_parameters() if v.grad is not None})
    loss2, n2 = crit(asm, (x.detach(), None), fx['in']['target2'])
    assert n2 == fx['out']['sample_size2']
    got['loss2'] = loss2.reshape(1)
    check_outputs(fx, got)


DEC_KW = dict(vocab_size=600, dim=64, heads=4, ffn=128, cutoff=(100, 300))


@pytest.mark.parametrize('kind', ['flattened', 'faces_objects', 'faces_parallel', 'flattened_no_image',
                                  'flattened_lightweight', 'flattened_prenorm'])
def test_decoder(golden, kind):
    fx = golden('decoder_' + kind)
    dec = build_decoder(kind, article_dim=64 if kind.startswith('flattened') else 1024, **DEC_KW).eval()
    load_sd(dec, fx['sd'])
    ins = fx['in']
    ctx = {k: v for k, v in ins.items() if k not in ('ids', 'target')}
    out = dec({'roberta': ins['ids']}, ctx)
    loss, n = AdaptiveLoss(1)(dec.adaptive_softmax, out, ins['target'])
    (loss / n).backward()
    assert n == fx['out']['sample_size']
    got = dict(x=out[0], loss=loss.reshape(1))
    pd = dict(dec.named_parameters())
    for k in list(fx['out']) + list(fx.get('sub', {})):
        if k.startswith('g_'):
            got[k] = pd[k[2:]].grad
        if k.startswith('attn0_'):
            got[k] = torch.from_numpy(out[1]['attn'][0][k[6:]])
    st = {}
    T = ins['ids'].shape[1]
    got['x_incremental'] = torch.cat(
        [dec({'roberta': ins['ids'][:, t:t + 1]}, ctx, incremental_state=st)[0] for t in range(T)], dim=1)
    check_outputs(fx, got)


def test_lstm_decoder(golden):
    """LSTMDecoder of the GloVe/LSTM baseline (decoder_flattened_lstm.py:68-208) vs the reference's outputs."""
    from oracle.build import build_embedder
    from oracle.lstm import LSTMDecoder
    fx = golden('decoder_lstm')
    dec = LSTMDecoder(build_embedder(600, 64, (100, 300)), num_layers=3, hidden_size=48, dropout=0.1, vocab_size=600,
                      adaptive_softmax_cutoff=(100, 300), article_embed_size=300, image_embed_size=2048).eval()
    load_sd(dec, fx['sd'])
    ins = fx['in']
    ctx = {k: v for k, v in ins.items() if k not in ('ids', 'target')}
    out = dec({'roberta': ins['ids']}, ctx)
    loss, n = AdaptiveLoss(1)(dec.adaptive_softmax, out, ins['target'])
    (loss / n).backward()
    assert n == fx['out']['sample_size']
    got = dict(x=out[0], loss=loss.reshape(1))
    pd = dict(dec.named_parameters())
    for k in list(fx['out']) + list(fx.get('sub', {})):
        if k.startswith('g_'):
            got[k] = pd[k[2:]].grad
    check_outputs(fx, got)


class _TableRoberta:
    """the fixture generator's stand-in article encoder (ref_import.StandInEncoders.Roberta)"""

    def __init__(self, dim, n_layers=25, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.tables = torch.randn(n_layers, 64, dim, generator=g) * 0.5

    def extract_features(self, ids, return_all_hiddens=False):
        outs = [t[ids % 64] for t in self.tables]
        return outs if return_all_hiddens else outs[-1]


class _PoolResnet(torch.nn.Module):
    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.proj = torch.randn(2048, 3, generator=g) * 0.3

    def forward(self, image):
        p = torch.nn.functional.avg_pool2d(image, 32)
        return torch.relu(torch.einsum('oc,bchw->bohw', self.proj, p))


@pytest.mark.parametrize('kind', ['flattened', 'faces_objects'])
def test_model_loss_and_greedy_generation(golden, kind):
    fx = golden('model_' + kind)
    art_dim = 64 if kind == 'flattened' else 1024
    model = build_model(kind, _PoolResnet(), _TableRoberta(art_dim), article_dim=art_dim,
                        **DEC_KW).eval()
    sd = {k: v for k, v in fx['sd'].items()}
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    ins = fx['in']

    def batch():
        b = dict(context={'roberta': ins['article_ids'].clone()}, image=ins['image'].clone(),
                 caption={'roberta': ins['caption_ids'].clone()})
        if kind == 'faces_objects':
            f, o = ins['face_embeds'].clone(), ins['obj_embeds'].clone()
            for i in range(f.shape[0]):
                f[i, int(ins['n_faces'][i]):] = float('nan')
                o[i, int(ins['n_objs'][i]):] = float('nan')
            b.update(face_embeds=f, obj_embeds=o)
        return b
    out = model(**batch())
    assert out['sample_size'] == fx['out']['sample_size']
    close(out['loss'].reshape(1), fx['out']['loss'])
    gen = model.generate(**batch())
    assert torch.equal(gen['gen_ids'], fx['out']['gen_ids'])          # bit-exact token ids
    close(gen['log_probs'], fx['out']['gen_log_probs'], atol=1e-4)


def test_baseline_glove_model(golden):
    """BaselineGloveModel + LSTMDecoder (expt/*/1_lstm_glove): loss and bit-exact greedy ids vs the reference."""
    from oracle.build import build_embedder
    from oracle.lstm import BaselineGloveModel, LSTMDecoder
    fx = golden('model_baseline_glove')
    dec = LSTMDecoder(build_embedder(600, 64, (100, 300)), num_layers=2, hidden_size=48, dropout=0.1, vocab_size=600,
                      adaptive_softmax_cutoff=(100, 300), article_embed_size=300, image_embed_size=2048)
    model = BaselineGloveModel(dec, AdaptiveLoss(1), _PoolResnet()).eval()
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in fx['sd'].items() if k in own}, strict=False)
    ins = fx['in']
    out = model(ins['image'].clone(), ins['caption'].clone(), ins['context_vectors'].clone())
    assert out['sample_size'] == fx['out']['sample_size']
    close(out['loss'].reshape(1), fx['out']['loss'])
    lp, ids = model.generate(ins['image'].clone(), ins['caption'].clone(), ins['context_vectors'].clone())
    assert torch.equal(ids, fx['out']['gen_ids'])
    close(lp, fx['out']['gen_log_probs'], atol=1e-4)


def test_transformer_glove_model(golden):
    """TransformerGloveModel (expt/*/2_transformer_glove): loss and bit-exact greedy ids vs the reference."""
    from oracle.lstm import TransformerGloveModel
    fx = golden('model_transformer_glove')
    dec = build_decoder('flattened', article_dim=300, **DEC_KW)
    model = TransformerGloveModel(dec, AdaptiveLoss(1), _PoolResnet()).eval()
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in fx['sd'].items() if k in own}, strict=False)
    ins = fx['in']
    out = model(ins['image'].clone(), ins['caption'].clone(), ins['context_vectors'].clone())
    assert out['sample_size'] == fx['out']['sample_size']
    close(out['loss'].reshape(1), fx['out']['loss'])
    lp, ids = model.generate(ins['image'].clone(), ins['caption'].clone(), ins['context_vectors'].clone())
    assert torch.equal(ids, fx['out']['gen_ids'])
    close(lp, fx['out']['gen_log_probs'], atol=1e-4)
