"""GPU: multi-tensor BertAdam kernel and whole optimisation steps vs the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
KW = dict(vocab_size=600, dim=64, heads=4, ffn=128, cutoff=(100, 300))


@pytest.fixture(autouse=True)
def _gpu():
    import tell_amd
    tell_amd.hip.require_gpu()
    yield
    torch.cuda.synchronize()


def test_bertadam_kernel_matches_oracle():
    from oracle.optim import BertAdam as OAdam
    from tell_amd.training.optimizers import BertAdam, FlatParams
    torch.manual_seed(0)
    shapes = [(300, 70), (5,), (1024,), (33, 1), (2049,)]
    cpu = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    gpu = [torch.nn.Parameter(p.detach().clone()) for p in cpu]
    flat = FlatParams([('p%d' % i, p) for i, p in enumerate(gpu)], DEV)
    cfg = dict(lr=1e-2, warmup=0.2, t_total=10, b1=0.9, b2=0.98, e=1e-6, weight_decay=1e-2, max_grad_norm=0.5)
    opt = BertAdam(flat, **cfg)
    ref = OAdam(cpu, **cfg)
    for step in range(4):
        for i, (a, b) in enumerate(zip(cpu, gpu)):
            g = torch.randn_like(a) * (0.001 if i == 1 else 1.0)      # tensor 1 stays below the clip threshold
            a.grad = g.clone()
            b.grad.copy_(g.to(DEV))
        ref.step()
        opt.step()
        for a, b in zip(cpu, gpu):
            torch.testing.assert_close(b.detach().cpu(), a.detach(), rtol=1e-5, atol=1e-6)
    assert opt.step_count == 4


class _Rob(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.register_buffer('tab', torch.randn(3, 64, dim, generator=g) * 0.5)

    def extract_features(self, ids, return_all_hiddens=False):
        import tell_amd
        out = self.tab[:, ids % 64]
        if out.is_cuda:
            return out.to(tell_amd.compute_dtype())
        return list(out)


class _Res(torch.nn.Module):
    def __init__(self, nhwc):
        super().__init__()
        g = torch.Generator().manual_seed(4)
        self.register_buffer('proj', torch.randn(2048, 3, generator=g) * 0.3)
        self.nhwc = nhwc

    def forward(self, image):
        import tell_amd
        f = torch.relu(torch.einsum('oc,bchw->bohw', self.proj, torch.nn.functional.avg_pool2d(image.float(), 32)))
        if not self.nhwc:
            return f
        return f.permute(0, 2, 3, 1).reshape(f.shape[0], 49, 2048).to(tell_amd.compute_dtype()).contiguous()


def _no_dropout(model):
    for m in model.modules():
        for a in ('dropout', 'input_dropout', 'relu_dropout', 'weight_dropout'):
            if isinstance(getattr(m, a, None), float):
                setattr(m, a, 0.0)


@pytest.mark.parametrize('kind', ['flattened', 'faces_objects'])
def test_two_training_steps_match_oracle_fp32(kind):
    """forward + loss + backward + BertAdam, twice, parameters compared after each step."""
    import tell_amd
    from oracle.build import build_model as obuild
    from oracle.optim import BertAdam as OAdam
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    adim = 64 if kind == 'flattened' else 1024
    gpu = build_model(kind, _Res(True), _Rob(adim), n_bert_layers=3, article_dim=adim, **KW)
    cpu = obuild(kind, _Res(False), _Rob(adim), n_bert_layers=3, article_dim=adim, **KW).train()
    cpu.load_state_dict({k: v for k, v in gpu.state_dict().items() if k in cpu.state_dict()}, strict=False)
    _no_dropout(gpu)
    _no_dropout(cpu)
    ocfg = dict(lr=5e-3, warmup=0.5, t_total=4, b1=0.9, b2=0.98, e=1e-6, weight_decay=1e-5, max_grad_norm=0.1)
    trainer = Trainer(gpu, dict(ocfg), device=DEV)
    ref_opt = OAdam([p for n, p in cpu.named_parameters() if not n.startswith(('resnet', 'roberta'))], **ocfg)
    for step in range(3):
        batch = synthetic_batch(B=3, article_len=20, caption_len=9, faces_objects=(kind == 'faces_objects'),
                                vocab=600, cutoffs=(100, 300), seed=50 + step, variable=True)
        ref_opt.zero_grad()
        ref = cpu(**{k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in batch.items()})
        ref['loss'].backward()
        ref_opt.step()
        dev_batch = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV))
                     for k, v in batch.items()}
        loss = trainer.train_one_batch(dev_batch)
        assert abs(float(loss) - float(ref['loss'])) <= 1e-3 * abs(float(ref['loss'])), (step, float(loss), float(ref['loss']))
    cp = dict(cpu.named_parameters())
    worst = 0.0
    for n, p in gpu.named_parameters():
        if n.startswith(('resnet', 'roberta')):
            continue
        d = (p.detach().cpu() - cp[n].detach()).abs().max().item()
        worst = max(worst, d / (cp[n].detach().abs().max().item() + 1e-6))
    assert worst < 2e-3, worst


def test_bf16_training_reduces_loss_with_dropout():
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.bfloat16)
    tell_amd.manual_seed(11)
    torch.manual_seed(1)
    model = build_model('faces_objects', _Res(True), _Rob(1024), n_bert_layers=3, **KW)
    trainer = Trainer(model, dict(lr=3e-3, warmup=0.1, t_total=60, max_grad_norm=1.0, weight_decay=0.0), device=DEV)
    batch = synthetic_batch(B=4, article_len=24, caption_len=12, faces_objects=True, vocab=600, cutoffs=(100, 300), seed=7)
    dev_batch = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV)) for k, v in batch.items()}
    losses = []
    for _ in range(40):
        b = {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in dev_batch.items()}
        losses.append(float(trainer.train_one_batch(b)))
    assert all(l == l for l in losses)                   # no NaN
    assert sum(losses[-5:]) / 5 < 0.6 * sum(losses[:3]) / 3, losses
    # the optimizer kernel keeps the bf16 working copy of every parameter and leaves the gradient buffer zeroed
    trainer.finish_update()
    torch.cuda.synchronize()
    # (zeroed but for the weight matrices whose single dense product stores over them: ops.py 'Gradient stores')
    assert trainer.flat.stored_numel > 0
    assert all(float(p.grad.abs().max()) == 0.0 for p in trainer.flat.params if not getattr(p, '_tell_grad_store', False))
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert torch.equal(p._tell_shadow, p.detach().to(torch.bfloat16)), n


@pytest.mark.parametrize('kind', ['flattened', 'faces_objects'])
def test_beam_search_matches_oracle_definition_fp32(kind):
    """beam search on the cached generator (replicated K/V, reordered DynamicConv buffers) against
    the plain prefix-re-decoding definition in oracle/beam.py; beam 1 through the beam code == greedy ids."""
    import tell_amd
    from oracle.beam import beam_search
    from oracle.build import build_model as obuild
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(5)
    adim = 64 if kind == 'flattened' else 1024
    gpu = build_model(kind, _Res(True), _Rob(adim), n_bert_layers=3, article_dim=adim, **KW)
    cpu = obuild(kind, _Res(False), _Rob(adim), n_bert_layers=3, article_dim=adim, **KW).eval()
    sd = gpu.state_dict()
    # sharpen the output distribution a little so that hypotheses finish at different lengths
    for k in sd:
        if k.endswith('adaptive_softmax.head.word_proj.weight') or k.endswith('embed_tokens.embeddings.0.weight'):
            sd[k][2] *= 1.6                                              # EOS row
    gpu.load_state_dict(sd)
    cpu.load_state_dict({k: v for k, v in sd.items() if k in cpu.state_dict()}, strict=False)
    gpu.eval().to(DEV)
    batch = synthetic_batch(B=3, article_len=20, caption_len=9, faces_objects=(kind == 'faces_objects'), vocab=600,
                            cutoffs=(100, 300), seed=77, variable=True)
    dev_batch = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV))
                 for k, v in batch.items()}
    clone = lambda b: {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}   # noqa: E731
    with torch.no_grad():
        cap_ids, _, ctx = cpu._forward(**{k: v for k, v in clone(batch).items()})
        cids, _, gctx = gpu._forward(**clone(dev_batch))
        for K in (4, 2):
            ref_ids, ref_score = beam_search(cpu, cap_ids, ctx, K, gen_len=24)
            lp, got, _ = gpu._generate(cids, gctx, beam_size=K, gen_len=24)
            got = got.cpu()
            n = min(got.shape[1], ref_ids.shape[1])
            assert torch.equal(got[:, :n], ref_ids[:, :n]), (K, got, ref_ids)
            assert (got[:, n:] == 1).all() and (ref_ids[:, n:] == 1).all()
            assert torch.allclose(lp.sum(1).cpu(), ref_score, rtol=1e-4, atol=2e-4)
        _, greedy, _ = gpu._generate(cids, gctx, gen_len=24)
        _, one, _ = gpu._generate_beam(cids, gctx, 1, gen_len=24)
        n = min(greedy.shape[1], one.shape[1])
        assert torch.equal(greedy.cpu()[:, :n], one.cpu()[:, :n])


@pytest.mark.parametrize('side_streams', [False, True])
def test_pipelined_steps_equal_plain_steps_fp32(side_streams):
    """Trainer.train_one_batch(batch, next_batch=...) launches the next batch's encoders underneath the current
    step; losses and parameters must equal the plain schedule (same kernels, no dropout).  side_streams: also with the
    opt-in weight-gradient stream (TELL_WGRAD_STREAM=1) and asynchronous update stream (TELL_ASYNC_UPDATE=1)."""
    import copy
    import tell_amd
    from tell_amd import ops
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(2)
    a = build_model('faces_objects', _Res(True), _Rob(1024), n_bert_layers=3, **KW)
    _no_dropout(a)
    b = copy.deepcopy(a)
    ocfg = dict(lr=5e-3, warmup=0.5, t_total=6, b1=0.9, b2=0.98, e=1e-6, weight_decay=1e-5, max_grad_norm=0.1)
    ta = Trainer(a, dict(ocfg), device=DEV, async_update=False)
    tb = Trainer(b, dict(ocfg), device=DEV, async_update=side_streams)
    assert tb.async_update == side_streams
    wg0 = ops._WGRAD['enabled']
    try:
        _pipelined_vs_plain(ta, tb, a, b, side_streams)
    finally:
        ops.join_wgrad_stream()
        ops._WGRAD['enabled'] = wg0


def _pipelined_vs_plain(ta, tb, a, b, side_streams):
    from tell_amd import ops
    from tell_amd.data import synthetic_batch
    batches = []
    for s in range(4):
        bt = synthetic_batch(B=3, article_len=20, caption_len=9, faces_objects=True, vocab=600, cutoffs=(100, 300),
                             seed=90 + s, variable=True)
        batches.append({k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV))
                        for k, v in bt.items()})
    clone = lambda x: {k: (dict(v) if isinstance(v, dict) else v) for k, v in x.items()}   # noqa: E731
    for s in range(4):
        ops._WGRAD['enabled'] = False
        la = ta.train_one_batch(clone(batches[s]))
        ops.join_wgrad_stream()
        ops._WGRAD['enabled'] = side_streams
        lb = tb.train_one_batch(clone(batches[s]), next_batch=batches[s + 1] if s + 1 < 4 else None)
        # not bit-exact: the embedding-table gradient accumulates duplicate tokens with fp32 atomics
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la)), (s, float(la), float(lb))
    tb.finish_update()
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert (p - q).norm() <= 1e-4 * (p.norm() + 1e-6), n


@pytest.mark.parametrize('kind', ['faces_parallel', 'flattened_no_image'])
def test_model_variants_train_step_matches_oracle_fp32(kind):
    """`transformer_faces` + `dynamic_conv_decoder_faces_parallel` (expt/*/8_transformer_faces) and
    `transformer_flattened` + `dynamic_conv_decoder_flattened_no_image` (expt/*/4_no_image) - one full step."""
    import tell_amd
    from oracle.build import build_model as obuild
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(3)
    adim = 1024 if kind == 'faces_parallel' else 64
    gpu = build_model(kind, _Res(True), _Rob(adim), n_bert_layers=3, article_dim=adim, **KW)
    cpu = obuild(kind, _Res(False), _Rob(adim), n_bert_layers=3, article_dim=adim, **KW).train()
    cpu.load_state_dict({k: v for k, v in gpu.state_dict().items() if k in cpu.state_dict()}, strict=False)
    _no_dropout(gpu)
    _no_dropout(cpu)
    gpu.to(DEV).train()
    batch = synthetic_batch(B=3, article_len=20, caption_len=9, faces_objects=(kind == 'faces_parallel'), vocab=600,
                            cutoffs=(100, 300), seed=31, variable=True)
    if kind == 'faces_parallel':
        batch.pop('obj_embeds')
    ref = cpu(**{k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in batch.items()})
    ref['loss'].backward()
    dev_batch = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV))
                 for k, v in batch.items()}
    out = gpu(**dev_batch)
    out['loss'].backward()
    tell_amd.ops.join_wgrad_stream()
    torch.cuda.synchronize()
    assert abs(float(out['loss']) - float(ref['loss'])) <= 1e-3 * abs(float(ref['loss']))
    cp = dict(cpu.named_parameters())
    checked = 0
    for n, p in gpu.named_parameters():
        if n.startswith(('resnet', 'roberta')) or p.grad is None or cp[n].grad is None:
            continue
        g, r = p.grad.detach().cpu(), cp[n].grad
        assert (g - r).norm() <= 2e-3 * (r.norm() + 1e-6), n
        checked += 1
    assert checked > 40


def test_baseline_glove_training_steps_match_oracle_fp32():
    """`baseline_glove` (LSTM decoder) through the Trainer - forward, loss, backward of every LSTM /
    attention kernel, BertAdam - twice; loss and every gradient of the first step against the CPU oracle, finite
    decreasing loss afterwards."""
    import tell_amd
    from oracle.build import build_embedder as obuild_embedder
    from oracle.lstm import BaselineGloveModel as OModel, LSTMDecoder as ODecoder
    from oracle.modules import AdaptiveLoss as OLoss
    from tell_amd.build import build_embedder
    from tell_amd.models import BaselineGloveModel, LSTMDecoder
    from tell_amd.modules import AdaptiveLoss
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(8)
    dec = LSTMDecoder(None, build_embedder(600, 64, (100, 300), 512), num_layers=2, hidden_size=48, dropout=0.0,
                      share_decoder_input_output_embed=True, vocab_size=600, adaptive_softmax_cutoff=[100, 300],
                      tie_adaptive_weights=True, adaptive_softmax_dropout=0, tie_adaptive_proj=False,
                      adaptive_softmax_factor=1, article_embed_size=300, image_embed_size=2048)
    gpu = BaselineGloveModel(None, dec, AdaptiveLoss(1), resnet=_Res(True))
    odec = ODecoder(obuild_embedder(600, 64, (100, 300)), num_layers=2, hidden_size=48, dropout=0.0, vocab_size=600,
                    adaptive_softmax_cutoff=(100, 300), article_embed_size=300, image_embed_size=2048)
    cpu = OModel(odec, OLoss(1), _Res(False)).train()
    cpu.load_state_dict({k: v for k, v in gpu.state_dict().items() if k in cpu.state_dict()}, strict=False)
    g = torch.Generator().manual_seed(9)
    B = 3
    image = torch.randn(B, 3, 224, 224, generator=g)
    cap = torch.randint(4, 600, (B, 9), generator=g)
    cap[:, 0] = 0
    cap[2, 6:] = 1
    cv = torch.randn(B, 12, 300, generator=g)
    cv[0, 9:] = float('nan')
    cv[2, 5:] = float('nan')
    ref = cpu(image.clone(), cap.clone(), cv.clone())
    ref['loss'].backward()
    trainer = Trainer(gpu, dict(lr=2e-3, warmup=0.5, t_total=6, max_grad_norm=0.1, weight_decay=0.0), device=DEV,
                      no_grad=(r'^resnet',))
    batch = lambda: dict(image=image.to(DEV), caption={'roberta': cap.to(DEV)}, context_vectors=cv.to(DEV))  # noqa: E731
    gpu.train()
    out = gpu(**batch())
    out['loss'].backward()
    torch.cuda.synchronize()
    assert abs(float(out['loss']) - float(ref['loss'])) <= 1e-3 * abs(float(ref['loss']))
    cp = dict(cpu.named_parameters())
    checked = 0
    for n, p in gpu.named_parameters():
        if n.startswith('resnet') or p.grad is None or cp[n].grad is None:
            continue
        gg, r = p.grad.detach().cpu(), cp[n].grad
        assert (gg - r).norm() <= 2e-3 * (r.norm() + 1e-6), n
        checked += 1
    assert checked >= 30, checked
    trainer.flat.zero_grad()
    losses = [float(trainer.train_one_batch(batch())) for _ in range(4)]
    assert all(l == l and abs(l) < 1e4 for l in losses) and losses[-1] < losses[0], losses


@pytest.mark.parametrize('weigh_bert', [False, True])
def test_lstm_decoder_behind_roberta_model_trains_and_generates(weigh_bert):
    """expt/*/3_lstm_roberta: `transformer_flattened` with `lstm_decoder_flattened` (article_embed_size 1024): a training
    step through the Trainer and a greedy decode (state-carrying generator) run end to end."""
    import tell_amd
    from tell_amd.build import build_embedder
    from tell_amd.data import synthetic_batch
    from tell_amd.models import LSTMDecoder, TransformerFlattenedModel
    from tell_amd.modules import AdaptiveLoss
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(11)
    dec = LSTMDecoder(None, build_embedder(600, 64, (100, 300), 512), num_layers=2, hidden_size=48, dropout=0.1,
                      share_decoder_input_output_embed=True, vocab_size=600, adaptive_softmax_cutoff=[100, 300],
                      tie_adaptive_weights=True, adaptive_softmax_dropout=0, tie_adaptive_proj=False,
                      adaptive_softmax_factor=1, article_embed_size=64, image_embed_size=2048)
    model = TransformerFlattenedModel(None, dec, AdaptiveLoss(1), weigh_bert=weigh_bert, vocab_size=600, resnet=_Res(True),
                                      roberta=_Rob(64), n_bert_layers=3)   # weigh_bert: the article states carry a gradient
    tr = Trainer(model, dict(lr=2e-3, warmup=0.5, t_total=8, max_grad_norm=0.1, weight_decay=0.0), device=DEV)
    bt = synthetic_batch(B=3, article_len=20, caption_len=9, vocab=600, cutoffs=(100, 300), seed=70)
    dev = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV)) for k, v in bt.items()}
    clone = lambda x: {k: (dict(v) if isinstance(v, dict) else v) for k, v in x.items()}   # noqa: E731
    losses = [float(tr.train_one_batch(clone(dev))) for _ in range(5)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    model.eval()
    gen = model.generate(**clone(dev))
    assert gen['gen_ids'].shape[0] == 3 and 2 <= gen['gen_ids'].shape[1] <= 101


def test_dp_code_path_one_rank_rccl(monkeypatch):
    """Every collective of the data-parallel step (token-count all-reduce, NaN flag, bf16-on-the-wire gradient
    all-reduce on the update stream) through a real 1-rank RCCL group; result == the non-DP trainer up to the
    bf16 rounding of the exchanged gradients."""
    import copy
    import torch.distributed as dist
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.bfloat16)
    tell_amd.manual_seed(4)
    torch.manual_seed(4)
    a = build_model('flattened', _Res(True), _Rob(64), n_bert_layers=3, article_dim=64, **KW)
    _no_dropout(a)
    b = copy.deepcopy(a)
    ocfg = dict(lr=2e-3, warmup=0.5, t_total=6, max_grad_norm=0.1, weight_decay=0.0)
    plain = Trainer(a, dict(ocfg), device=DEV)
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', '29581')
    monkeypatch.setenv('TELL_DP_SELFTEST', '1')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        dpt = Trainer(b, dict(ocfg), device=DEV, nan_check=True)
        assert dpt.dp and dpt.allreduce_dtype == torch.bfloat16
        for s in range(3):
            bt = synthetic_batch(B=3, article_len=20, caption_len=9, vocab=600, cutoffs=(100, 300), seed=40 + s)
            dev = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV))
                   for k, v in bt.items()}
            clone = lambda x: {k: (dict(v) if isinstance(v, dict) else v) for k, v in x.items()}   # noqa: E731
            l0, l1 = plain.train_one_batch(clone(dev)), dpt.train_one_batch(clone(dev))
            assert abs(float(l0) - float(l1)) <= 2e-2 * abs(float(l0)), (s, float(l0), float(l1))
        dpt.finish_update()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
        from tell_amd import runtime as rt_
        rt_.set_grad_ready_callback(None)
    # BertAdam normalises every update by sqrt(v): elements whose gradient is ~0 can move by +-lr in either
    # direction after the bf16 rounding, so compare in the large (whole-model norm), not per tensor
    num = sum(float((p - q).norm() ** 2) for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters())
              if p.requires_grad)
    den = sum(float(p.norm() ** 2) for n, p in a.named_parameters() if p.requires_grad)
    assert (num / den) ** 0.5 < 1e-2, (num / den) ** 0.5


def test_dp_bucketed_allreduce_during_backward_one_rank_rccl(monkeypatch):
    """The gradient slices of the decoder layers are all-reduced while backward is still running.  A slice exchanged
    before its last contribution would be wrong on N > 1 ranks only, so the test emulates two identical ranks: every
    all-reduce is followed by x2 on the range it covered.  The bucketed schedule must then give the same weights as
    the single exchange after backward (any gradient added after its slice was reduced would count once, not twice)."""
    import copy
    import torch.distributed as dist
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.bfloat16)
    tell_amd.manual_seed(5)
    torch.manual_seed(5)
    a = build_model('flattened', _Res(True), _Rob(64), n_bert_layers=3, article_dim=64, **KW)
    _no_dropout(a)
    b = copy.deepcopy(a)
    ocfg = dict(lr=2e-3, warmup=0.5, t_total=6, max_grad_norm=0.1, weight_decay=0.0)
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', '29582')
    monkeypatch.setenv('TELL_DP_SELFTEST', '1')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        monkeypatch.setenv('TELL_DP_BUCKETED', '0')
        single = Trainer(a, dict(ocfg), device=DEV)
        monkeypatch.setenv('TELL_DP_BUCKETED', '1')
        bucketed = Trainer(b, dict(ocfg), device=DEV)
        assert not single._ranges and len(bucketed._ranges) == len(b.decoder.layers)
        single._test_reduce_scale = bucketed._test_reduce_scale = 2.0
        for s in range(3):
            bt = synthetic_batch(B=3, article_len=20, caption_len=9, vocab=600, cutoffs=(100, 300), seed=50 + s)
            dev = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV))
                   for k, v in bt.items()}
            clone = lambda x: {k: (dict(v) if isinstance(v, dict) else v) for k, v in x.items()}   # noqa: E731
            l0, l1 = single.train_one_batch(clone(dev)), bucketed.train_one_batch(clone(dev))
            assert abs(float(l0) - float(l1)) <= 1e-3 * abs(float(l0)), (s, float(l0), float(l1))
        assert bucketed.bucketed_reduces == 3 * len(b.decoder.layers) and single.bucketed_reduces == 0
        single.finish_update()
        bucketed.finish_update()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
        from tell_amd import runtime as rt
        rt.set_grad_ready_callback(None)
    num = sum(float((p - q).norm() ** 2) for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters())
              if p.requires_grad)
    den = sum(float(p.norm() ** 2) for n, p in a.named_parameters() if p.requires_grad)
    assert (num / den) ** 0.5 < 1e-4, (num / den) ** 0.5


def test_dp_two_real_ranks_share_one_gpu():
    """Two data-parallel ranks (gloo carries the gradients through the host, both ranks compute on cuda:0) with different,
    ragged batches: the per-layer exchange started during backward must leave both ranks with bit-identical weights
    after every step (tools/dp_two_ranks.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TELL_DP_BUCKETED='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29653', os.path.join(root, 'tools', 'dp_two_ranks.py')],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert 'RESULT ranks consistent' in out, out[-3000:]
    assert 'bucketed reduces so far: 16' in out, out[-3000:]


def test_pipelined_steps_with_graphed_encoders_equal_eager_fp32():
    """Real (small) ResNet + RoBERTa encoders replayed as hipGraphs AND prefetched one batch ahead: the decoder step
    of batch N must see batch N's features although the graphs for batch N+1 are already running (the captures are
    double-buffered).  Reference: the same steps with graphs off and no prefetch."""
    import copy
    import tell_amd
    from tell_amd import graphs
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.models.resnet import ResNetFeatureExtractor
    from tell_amd.models.roberta import RobertaEncoder
    from tell_amd.training import Trainer
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(6)
    res = ResNetFeatureExtractor((1, 1, 1, 1), width=64)
    rob = RobertaEncoder(vocab=600, dim=64, ffn=128, layers=2, heads=1, max_positions=40, dropout=0.0,
                         attention_dropout=0.0)
    a = build_model('flattened', res, rob, n_bert_layers=3, article_dim=64, **KW)
    _no_dropout(a)
    b = copy.deepcopy(a)
    ocfg = dict(lr=5e-3, warmup=0.5, t_total=8, b1=0.9, b2=0.98, e=1e-6, weight_decay=1e-5, max_grad_norm=0.1)
    ta, tb = Trainer(a, dict(ocfg), device=DEV), Trainer(b, dict(ocfg), device=DEV)
    batches = []
    for s in range(5):
        bt = synthetic_batch(B=2, article_len=24, caption_len=9, vocab=600, cutoffs=(100, 300), seed=70 + s)
        bt['image'] = bt['image'][:, :, :64, :64].contiguous()        # 2x2 regions instead of 7x7: a quick trunk
        batches.append({k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV))
                        for k, v in bt.items()})
    clone = lambda x: {k: (dict(v) if isinstance(v, dict) else v) for k, v in x.items()}   # noqa: E731
    was = graphs.ENABLED
    try:
        for s in range(5):
            graphs.ENABLED = False
            la = ta.train_one_batch(clone(batches[s]))
            graphs.ENABLED = True
            lb = tb.train_one_batch(clone(batches[s]), next_batch=batches[s + 1] if s + 1 < 5 else None)
            assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la)), (s, float(la), float(lb))
        states = [v['state'] for v in b.__dict__['_roberta_graph'].entries.values()]
        assert states == ['ready'], states
    finally:
        graphs.ENABLED = was
        tb.finish_update()
        torch.cuda.synchronize()


def test_bench_two_ranks_through_the_real_launcher():
    """`python bench.py --gpus 2` end to end - the launcher (torch.distributed.run, rendezvous on 127.0.0.1), two ranks of
    the configs[2] step with the gradient exchange in every leg, max-over-ranks timing, the per-rank report - rehearsed on
    ONE GPU (TELL_BENCH_ONE_GPU=1: every rank computes on cuda:0; gradients over gloo, because RCCL refuses two ranks on
    one device).  The first real 8-GPU run must not be the first run of this code path: every rank has to enter every leg
    (a rank that skips one deadlocks the exchange - round 3), the line has to carry the exchange's cost, and the replicas
    have to hold bit-identical weights at the end."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TELL_BENCH_ONE_GPU='1', TELL_DP_BACKEND='gloo', OMP_NUM_THREADS='4')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TELL_DP_SELFTEST'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--windows', '2',
           '--burn-in', '0', '--roofline-steps', '1', '--no-pmc']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]     # rank 0 alone prints
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['warmup'] == 2 and j['scaling'] == 'weak'
    assert j['config']['global_batch'] == 64 and j['config']['parallelism'] == 'dp2'
    assert j['windows'] == 2 and len(j['windows_ms_per_step']) == 2 and j['value'] > 0
    # value = the whole job: both ranks' samples over the slowest rank's window
    assert abs(j['value'] - 2 * 32 * 3 / (j['ms_per_step'] * 3e-3)) <= 0.01 * j['value']
    dp = j['dp']
    assert dp['allreduce_ms'] > 0 and dp['exposed_allreduce_ms'] >= 0 and dp['gradient_mbytes'] > 100
    assert dp['every_rank_ran_the_same_legs'] and dp['weights_identical_across_ranks'], dp
    assert [r['rank'] for r in dp['ranks']] == [0, 1]
    assert dp['ranks'][0]['legs'] == ['timed_region', 'roofline_leg'], dp['ranks']
    assert j['config']['skipped_steps'] == 0 and j['roofline']['frac'] > 0
    assert 'cpu_baseline' not in j and 'generation' not in j          # single-GPU legs stay out of a multi-rank line
