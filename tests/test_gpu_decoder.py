"""GPU parity of the full decoders / caption models against fixtures produced by the REAL
reference (tests/golden).  fp32 mode: loss/logits within 1e-3 relative, greedy token ids
bit-exact (BASELINE.json north_star); bf16 mode: bf16-level tolerance."""
import pytest
import torch

import seeded

pytestmark = pytest.mark.gpu
DEV = 'cuda'
DTYPES = [torch.float32, torch.bfloat16]
DEC_KW = dict(vocab_size=600, dim=64, heads=4, ffn=128, cutoff=(100, 300))


@pytest.fixture(autouse=True)
def _gpu():
    import tell_amd
    tell_amd.hip.require_gpu()
    yield
    torch.cuda.synchronize()


def close(a, b, dtype, scale=1.0, **kw):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    if dtype == torch.float32:
        t = dict(rtol=1e-3, atol=2e-5)
        t['atol'] *= scale
        t.update(kw)
        torch.testing.assert_close(a, b, **t)
    else:
        assert a.shape == b.shape
        rel = (a - b).norm().item() / (b.norm().item() + 1e-6)
        assert rel < kw.get('rtol', 5e-2), 'relative error %.4g' % rel


def _ctx_to_dev(ins, dtype):
    ctx = {}
    for k, v in ins.items():
        if k in ('ids', 'target'):
            continue
        ctx[k] = v.to(DEV) if v.dtype == torch.bool else v.to(DEV, dtype)
    return ctx


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('kind', ['flattened', 'faces_objects', 'faces_parallel', 'flattened_no_image',
                                  'flattened_lightweight', 'flattened_prenorm'])
def test_decoder_golden(golden, dtype, kind):
    import tell_amd
    from tell_amd.build import build_decoder
    from tell_amd.modules import AdaptiveLoss
    tell_amd.set_compute_dtype(dtype)
    fx = golden('decoder_' + kind)
    dec = build_decoder(kind, article_dim=64 if kind.startswith('flattened') else 1024, **DEC_KW).eval()
    dec.load_state_dict(fx['sd'], strict=False)
    dec.to(DEV)
    ins = fx['in']
    ctx = _ctx_to_dev(ins, dtype)
    torch.set_grad_enabled(True)
    dec.train()
    for m in dec.modules():                       # training graph, but every dropout off (reference ran eval())
        for attr in ('dropout', 'input_dropout', 'relu_dropout', 'weight_dropout'):
            if hasattr(m, attr) and isinstance(getattr(m, attr), float):
                setattr(m, attr, 0.0)
    out = dec({'roberta': ins['ids'].to(DEV)}, ctx)
    loss, n = AdaptiveLoss(1)(dec.adaptive_softmax, out, ins['target'].to(DEV))
    (loss / n.float()).sum().backward()
    assert int(n) == fx['out']['sample_size']
    close(out[0], fx['out']['x'], dtype, scale=20)
    close(loss.reshape(1), fx['out']['loss'], dtype, rtol=1e-3 if dtype == torch.float32 else 3e-2)
    pd = dict(dec.named_parameters())
    # bf16: gradients of an 18-token, 4-layer stack accumulate bf16 rounding; 15 % in norm
    gtol = dict(rtol=0.15) if dtype == torch.bfloat16 else {}
    for k, v in fx['out'].items():
        if k.startswith('g_'):
            close(pd[k[2:]].grad, v, dtype, scale=10, **gtol)
    for k, v in fx.get('sub', {}).items():
        if k.startswith('g_'):
            close(torch.from_numpy(seeded.subsample(pd[k[2:]].grad.float().cpu().numpy())), v, dtype, scale=10,
                  **gtol)
    # incremental (generation) path == full path, and attention weights of layer 0
    dec.eval()
    for layer in dec.layers:
        layer.need_attn = True
    with torch.no_grad():
        st = {}
        ids = ins['ids'].to(DEV)
        inc = torch.cat([dec({'roberta': ids[:, t:t + 1]}, ctx, incremental_state=st)[0]
                         for t in range(ids.shape[1])], dim=1)
        close(inc, fx['out']['x_incremental'], dtype, scale=20)
        full = dec({'roberta': ids}, ctx)
        for k, v in fx['out'].items():
            if k.startswith('attn0_'):
                close(torch.from_numpy(full[1]['attn'][0][k[6:]]), v, dtype)


class _TableRoberta(torch.nn.Module):
    """Test-only stand-in article encoder (same tables as tests/golden/ref_import.py)."""

    def __init__(self, dim, n_layers=25, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer('tables', torch.randn(n_layers, 64, dim, generator=g) * 0.5)

    def extract_features(self, ids, return_all_hiddens=False):
        import tell_amd
        out = self.tables[:, ids % 64].to(tell_amd.compute_dtype())      # [L,B,S,E]
        return out if return_all_hiddens else out[-1]


class _PoolResnet(torch.nn.Module):
    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer('proj', torch.randn(2048, 3, generator=g) * 0.3)

    def forward(self, image):
        import tell_amd
        p = torch.nn.functional.avg_pool2d(image.float(), 32)
        f = torch.relu(torch.einsum('oc,bchw->bohw', self.proj, p))
        B = f.shape[0]
        return f.permute(0, 2, 3, 1).reshape(B, 49, 2048).to(tell_amd.compute_dtype()).contiguous()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('kind', ['flattened', 'faces_objects'])
def test_model_loss_and_greedy_generation_golden(golden, dtype, kind):
    import tell_amd
    from tell_amd.build import build_model
    tell_amd.set_compute_dtype(dtype)
    fx = golden('model_' + kind)
    art_dim = 64 if kind == 'flattened' else 1024
    model = build_model(kind, _PoolResnet(), _TableRoberta(art_dim), article_dim=art_dim, **DEC_KW).eval()
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in fx['sd'].items() if k in own}, strict=False)
    model.to(DEV)
    ins = fx['in']

    def batch():
        b = dict(context={'roberta': ins['article_ids'].to(DEV)}, image=ins['image'].to(DEV),
                 caption={'roberta': ins['caption_ids'].to(DEV)})
        if kind == 'faces_objects':
            f, o = ins['face_embeds'].clone(), ins['obj_embeds'].clone()
            for i in range(f.shape[0]):
                f[i, int(ins['n_faces'][i]):] = float('nan')
                o[i, int(ins['n_objs'][i]):] = float('nan')
            b.update(face_embeds=f.to(DEV), obj_embeds=o.to(DEV))
        return b
    with torch.no_grad():
        out = model(**batch())
    assert int(out['sample_size']) == fx['out']['sample_size']
    close(out['loss'].reshape(1), fx['out']['loss'], dtype, rtol=1e-3 if dtype == torch.float32 else 3e-2)
    ref_ids = fx['out']['gen_ids']
    # K/V-cached static-batch generator (1st call: eager step 0 + capture at step 1; 2nd call: every step is a
    # replay of the captured decode step, position offset from the device counter) and the reference's control flow
    for fast in (True, True, False):
        model.fast_generation = fast
        gen = model.generate(**batch())
        got = gen['gen_ids'].cpu()
        if dtype == torch.float32:
            assert got.shape == ref_ids.shape and torch.equal(got, ref_ids), fast    # bit-exact greedy token ids
            if fast and tell_amd.graphs.ENABLED:             # (TELL_GRAPHS=0 runs the same steps eagerly)
                hs = list(model.__dict__.get('_decode_graphs', {}).values())
                assert hs and all(h['graph'] not in (None, False) for h in hs), [h.get('error') for h in hs]
            close(gen['log_probs'], fx['out']['gen_log_probs'], dtype, atol=2e-4)
        else:
            # bf16: a random-init model's arg-max has near-ties, and a free-running decode that takes one of them the other
            # way never comes back - so the gate is a MEASURED yardstick, not a constant: the CPU oracle under
            # torch.autocast(bfloat16) decodes the same batch (0.82 / 0.88 position-wise agreement with the fp32 reference
            # on these fixtures: one of the four rows leaves it); the HIP path may lose at most one row's worth more.
            n = min(got.shape[1], ref_ids.shape[1])
            agree = (got[:, :n] == ref_ids[:, :n]).float().mean().item()
            yard = _autocast_yardstick(kind, fx)
            assert agree >= yard['free_running'] - 1.0 / ref_ids.shape[0] - 1e-6, (fast, agree, yard)
    if dtype == torch.bfloat16:
        # ... and TEACHER-FORCED on the reference's own tokens (no divergence: every position is judged on its own): the
        # arg-max of the HIP decoder against the reference token, non-padding positions, within 2 % of the same figure for
        # the autocast oracle
        ref = ref_ids.to(DEV)
        with torch.no_grad():
            b = batch()
            _, _, ctx = model._forward(b['context'], b['image'], b['caption'], b.get('face_embeds'), b.get('obj_embeds'))
            out = model.decoder({'roberta': ref[:, :-1].contiguous()}, ctx)
            pred = model.decoder.get_normalized_probs(out, log_probs=True).argmax(-1).cpu()
        valid = ref_ids[:, 1:] != 1
        tf = (pred == ref_ids[:, 1:])[valid].float().mean().item()
        yard = _autocast_yardstick(kind, fx)
        print('\nbf16 greedy, %s: teacher-forced agreement %.4f (autocast oracle %.4f), free-running %.4f (%.4f)'
              % (kind, tf, yard['teacher_forced'], agree, yard['free_running']))
        assert tf >= yard['teacher_forced'] - 0.02, (tf, yard)


_YARD = {}


def _autocast_yardstick(kind, fx):
    """The CPU oracle (fp32 restatement of the reference) under torch.autocast('cpu', bfloat16) on the fixture's batch:
    position-wise agreement of its free-running greedy decode, and of its teacher-forced arg-max, with the fp32
    reference's token ids - what bf16 arithmetic alone costs on this model."""
    if kind in _YARD:
        return _YARD[kind]
    from oracle.build import build_model as obuild
    from test_oracle_golden import _PoolResnet as OResnet, _TableRoberta as ORoberta
    art_dim = 64 if kind == 'flattened' else 1024
    cpu = obuild(kind, OResnet(), ORoberta(art_dim), article_dim=art_dim, **DEC_KW).eval()
    own = cpu.state_dict()
    cpu.load_state_dict({k: v for k, v in fx['sd'].items() if k in own}, strict=False)
    ins, ref = fx['in'], fx['out']['gen_ids']

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
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        got = cpu.generate(**batch())['gen_ids']
        b = batch()
        _, _, ctx = cpu._forward(b['context'], b['image'], b['caption'], b.get('face_embeds'), b.get('obj_embeds'))
        out = cpu.decoder({'roberta': ref[:, :-1]}, ctx)
        pred = cpu.decoder.get_normalized_probs((out[0], None), log_probs=True).argmax(-1)
    n = min(got.shape[1], ref.shape[1])
    valid = ref[:, 1:] != 1
    _YARD[kind] = dict(free_running=(got[:, :n] == ref[:, :n]).float().mean().item(),
                       teacher_forced=(pred == ref[:, 1:])[valid].float().mean().item())
    return _YARD[kind]


@pytest.mark.parametrize('dtype', DTYPES)
def test_lstm_decoder_golden(golden, dtype):
    """`lstm_decoder_flattened` (the decoder of the GloVe/LSTM baseline): loss, outputs and gradients vs
    what the REFERENCE's LSTMDecoder produced."""
    import tell_amd
    from tell_amd.build import build_embedder
    from tell_amd.models import LSTMDecoder
    from tell_amd.modules import AdaptiveLoss
    tell_amd.set_compute_dtype(dtype)
    fx = golden('decoder_lstm')
    dec = LSTMDecoder(None, build_embedder(600, 64, (100, 300), 512), num_layers=3, hidden_size=48, dropout=0.1,
                      share_decoder_input_output_embed=True, vocab_size=600, adaptive_softmax_cutoff=[100, 300],
                      tie_adaptive_weights=True, adaptive_softmax_dropout=0, tie_adaptive_proj=False,
                      adaptive_softmax_factor=1, article_embed_size=300, image_embed_size=2048)
    dec.load_state_dict(fx['sd'], strict=False)
    dec.to(DEV)
    ins = fx['in']
    ctx = _ctx_to_dev(ins, dtype)
    torch.set_grad_enabled(True)
    dec.train()
    dec.dropout = 0.0                                   # training graph, dropout off (the reference ran eval())
    out = dec({'roberta': ins['ids'].to(DEV)}, ctx)
    loss, n = AdaptiveLoss(1)(dec.adaptive_softmax, out, ins['target'].to(DEV))
    (loss / n.float()).sum().backward()
    assert int(n) == fx['out']['sample_size']
    close(out[0], fx['out']['x'], dtype, scale=20)
    close(loss.reshape(1), fx['out']['loss'], dtype, rtol=1e-3 if dtype == torch.float32 else 3e-2)
    pd = dict(dec.named_parameters())
    gtol = dict(rtol=0.15) if dtype == torch.bfloat16 else {}
    for k, v in fx['out'].items():
        if k.startswith('g_'):
            close(pd[k[2:]].grad, v, dtype, scale=10, **gtol)
    for k, v in fx.get('sub', {}).items():
        if k.startswith('g_'):
            close(torch.from_numpy(seeded.subsample(pd[k[2:]].grad.float().cpu().numpy())), v, dtype, scale=10,
                  **gtol)


@pytest.mark.parametrize('dtype', DTYPES)
def test_baseline_glove_model_golden(golden, dtype):
    """`baseline_glove` (ResNet regions + GloVe article vectors -> LSTM decoder): loss and greedy token
    ids vs the REFERENCE; generation carries the LSTM state instead of re-decoding the prefix."""
    import tell_amd
    from tell_amd.build import build_embedder
    from tell_amd.models import BaselineGloveModel, LSTMDecoder
    from tell_amd.modules import AdaptiveLoss
    tell_amd.set_compute_dtype(dtype)
    fx = golden('model_baseline_glove')
    dec = LSTMDecoder(None, build_embedder(600, 64, (100, 300), 512), num_layers=2, hidden_size=48, dropout=0.1,
                      share_decoder_input_output_embed=True, vocab_size=600, adaptive_softmax_cutoff=[100, 300],
                      tie_adaptive_weights=True, adaptive_softmax_dropout=0, tie_adaptive_proj=False,
                      adaptive_softmax_factor=1, article_embed_size=300, image_embed_size=2048)
    model = BaselineGloveModel(None, dec, AdaptiveLoss(1), resnet=_PoolResnet()).eval()
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in fx['sd'].items() if k in own}, strict=False)
    model.to(DEV)
    ins = fx['in']
    batch = lambda: dict(image=ins['image'].to(DEV), caption={'roberta': ins['caption'].to(DEV)},   # noqa: E731
                         context_vectors=ins['context_vectors'].to(DEV))
    with torch.no_grad():
        out = model(**batch())
    assert int(out['sample_size']) == fx['out']['sample_size']
    close(out['loss'].reshape(1), fx['out']['loss'], dtype, rtol=1e-3 if dtype == torch.float32 else 3e-2)
    gen = model.generate(**batch())
    got, ref_ids = gen['gen_ids'].cpu(), fx['out']['gen_ids']
    if dtype == torch.float32:
        assert got.shape == ref_ids.shape and torch.equal(got, ref_ids)                  # bit-exact greedy token ids
        close(gen['log_probs'], fx['out']['gen_log_probs'], dtype, atol=2e-4)
    else:
        n = min(got.shape[1], ref_ids.shape[1])
        assert (got[:, :n] == ref_ids[:, :n]).float().mean().item() > 0.5


@pytest.mark.parametrize('dtype', DTYPES)
def test_transformer_glove_model_golden(golden, dtype):
    """`transformer_glove` (expt/*/2_transformer_glove): the flattened DynamicConv decoder over ResNet regions + GloVe
    vectors; loss and greedy ids (cached static-batch generator) vs the REFERENCE."""
    import tell_amd
    from tell_amd.build import build_decoder
    from tell_amd.models import TransformerGloveModel
    from tell_amd.modules import AdaptiveLoss
    tell_amd.set_compute_dtype(dtype)
    fx = golden('model_transformer_glove')
    dec = build_decoder('flattened', article_dim=300, **DEC_KW)
    model = TransformerGloveModel(None, dec, AdaptiveLoss(1), vocab_size=600, resnet=_PoolResnet()).eval()
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in fx['sd'].items() if k in own}, strict=False)
    model.to(DEV)
    ins = fx['in']
    batch = lambda: dict(image=ins['image'].to(DEV), caption={'roberta': ins['caption'].to(DEV)},   # noqa: E731
                         context_vectors=ins['context_vectors'].to(DEV))
    with torch.no_grad():
        out = model(**batch())
    assert int(out['sample_size']) == fx['out']['sample_size']
    close(out['loss'].reshape(1), fx['out']['loss'], dtype, rtol=1e-3 if dtype == torch.float32 else 3e-2)
    ref_ids = fx['out']['gen_ids']
    for fast in (True, False):
        model.fast_generation = fast
        gen = model.generate(**batch())
        got = gen['gen_ids'].cpu()
        if dtype == torch.float32:
            assert got.shape == ref_ids.shape and torch.equal(got, ref_ids), fast
            close(gen['log_probs'], fx['out']['gen_log_probs'], dtype, atol=2e-4)
        else:
            n = min(got.shape[1], ref_ids.shape[1])
            assert (got[:, :n] == ref_ids[:, :n]).float().mean().item() > 0.5
