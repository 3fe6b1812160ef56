"""GPU parity at the FULL size the benchmark times (E = 1024, 16 heads of 64, FFN 4096, V = 50265 with cutoffs
5000 / 20000, K = 3/7/15/31, contexts S = 512 / 49 / 4 / 64) against the CPU oracle - the kernels the bench actually
runs (direct-to-LDS 128x128 / 256x256 GEMMs, K-major wgrad / dgrad forms, MFMA attention, the 30 265-wide tail CE):
whole 4-layer decoders (loss, output, EVERY gradient tensor), full-vocabulary arg-max, full-size greedy decode.
fp32: loss / outputs within 1e-3 relative, greedy ids bit-exact (BASELINE.json north_star).  bf16: per-tensor
relative Frobenius error bounded by what bf16 rounding of a 4-layer, K <= 16384 accumulation chain gives (see BF16_*)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
B, T = 4, 32
SHAPES = {'image': (49, 2048), 'article': (512, 1024), 'faces': (4, 512), 'obj': (64, 2048)}

# bf16 bounds.  The yardstick is MEASURED, not guessed: the same oracle run under torch.autocast('cpu', bfloat16)
# (bf16 GEMM operands, fp32 accumulation / softmax / LayerNorm - the arithmetic class of the HIP bf16 path) against
# its own fp32 run.  At this size (random init, 4 layers, cancellation in the softmax / LayerNorm backward of nearly
# uniform distributions) that gives 4 % median and 7-8 % worst-tensor relative gradient error (embedding projections,
# layer-0 tap projection) - bf16 noise of the MODEL.  Every gradient tensor of the HIP path must stay within
# BF16_VS_AUTOCAST x the autocast error of the SAME tensor (+ BF16_FLOOR), and below the absolute ceiling BF16_CEIL:
# a kernel bug (a dropped term, a wrong tile) shows up as a tensor far outside its own yardstick.
BF16_LOSS = 3e-3
BF16_OUT = 1.5e-2
BF16_VS_AUTOCAST = 1.3
BF16_FLOOR = 5e-3
BF16_CEIL = 0.10


@pytest.fixture(autouse=True)
def _gpu():
    import tell_amd
    tell_amd.hip.require_gpu()
    yield
    torch.cuda.synchronize()


def _inputs(kind, seed=0):
    g = torch.Generator().manual_seed(seed)
    names = ['image', 'article'] + (['faces', 'obj'] if kind == 'faces_objects' else [])
    ctx = {}
    for n in names:
        S, C = SHAPES[n]
        x = torch.randn(S, B, C, generator=g) * 0.5
        if n in ('image', 'obj'):
            x = x.abs()
        lens = torch.randint(max(S // 2, 1), S + 1, (B,), generator=g)
        if n == 'image':
            lens = torch.full((B,), S)
        if n == 'faces':
            lens = torch.tensor([0, 1, 4, 2])                     # one sample without any face
        mask = torch.arange(S)[None, :] >= lens[:, None]
        x = x * (~mask).t()[:, :, None]                            # the reference zeroes padded rows (:375,379)
        ctx[n], ctx[n + '_mask'] = x, mask
    from tell_amd.data.synthetic import _ids
    lens = torch.tensor([T + 1, T - 3, T + 1, 9])
    cap = _ids(g, B, T + 1, lens, 50265, (5000, 20000))
    return ctx, cap[:, :-1].contiguous(), cap[:, 1:].contiguous()


def _no_dropout(m):
    for mod in m.modules():
        for a in ('dropout', 'input_dropout', 'relu_dropout', 'weight_dropout', 'attention_dropout'):
            if isinstance(getattr(mod, a, None), float):
                setattr(mod, a, 0.0)


_ORACLE = {}


def _oracle(kind):
    """Oracle decoder + its loss / output / gradients on the shared inputs (computed once per kind, CPU fp32)."""
    if kind not in _ORACLE:
        from oracle.build import build_decoder as obuild
        from oracle.modules import AdaptiveLoss as OLoss
        from tell_amd.build import build_decoder
        torch.manual_seed(0)
        dec = build_decoder(kind)                                  # the HIP-side module supplies the initial weights
        sd = {k: v.clone() for k, v in dec.state_dict().items()}
        ref = obuild(kind).train()
        ref.load_state_dict({k: v for k, v in sd.items() if k in ref.state_dict()}, strict=False)
        _no_dropout(ref)
        ctx, ids, tgt = _inputs(kind)
        torch.set_num_threads(min(32, torch.get_num_threads()))
        out = ref({'roberta': ids}, {k: v.clone() for k, v in ctx.items()})
        loss, n = OLoss(1)(ref.adaptive_softmax, out, tgt)
        (loss / n).backward()
        grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None}
        # the bf16 yardstick: the same module under CPU autocast
        ref.zero_grad(set_to_none=True)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            out16 = ref({'roberta': ids}, {k: v.clone() for k, v in ctx.items()})
            loss16, _ = OLoss(1)(ref.adaptive_softmax, out16, tgt)
        (loss16.float() / n).backward()
        yard = {k: _rel(p.grad, grads[k]) for k, p in ref.named_parameters()
                if p.grad is not None and grads[k].norm().item() >= 1e-12}
        ref.zero_grad(set_to_none=True)
        _ORACLE[kind] = dict(sd=sd, x=out[0].detach(), loss=float(loss), n=int(n), grads=grads, ref=ref,
                             inputs=(ctx, ids, tgt), yard=yard)
    return _ORACLE[kind]


def _to_dev(ctx, dtype):
    return {k: (v.to(DEV) if v.dtype == torch.bool else v.to(DEV, dtype)) for k, v in ctx.items()}


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('kind', ['flattened', 'faces_objects'])
def test_full_size_decoder_loss_output_and_every_gradient(kind, dtype):
    import tell_amd
    from tell_amd.build import build_decoder
    from tell_amd.modules import AdaptiveLoss
    o = _oracle(kind)
    tell_amd.set_compute_dtype(dtype)
    dec = build_decoder(kind)
    dec.load_state_dict(o['sd'])
    dec.to(DEV).train()
    _no_dropout(dec)
    ctx, ids, tgt = o['inputs']
    out = dec({'roberta': ids.to(DEV)}, _to_dev(ctx, dtype))
    loss, n = AdaptiveLoss(1)(dec.adaptive_softmax, out, tgt.to(DEV))
    (loss / n.float()).sum().backward()
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert int(n) == o['n']
    assert abs(float(loss) - o['loss']) <= (1e-4 if f32 else BF16_LOSS) * abs(o['loss']), (float(loss), o['loss'])
    r = _rel(out[0], o['x'])
    assert r <= (1e-4 if f32 else BF16_OUT), 'decoder output: relative error %.3e' % r
    worst, report = 0.0, []
    pd = dict(dec.named_parameters())
    assert set(o['grads']) <= set(pd)
    for k, gref in o['grads'].items():
        g = pd[k].grad
        assert g is not None, k
        if gref.norm().item() < 1e-12:                             # rows never touched (zero gradient on both sides)
            assert g.float().abs().max().item() <= 1e-6, k
            continue
        r = _rel(g, gref)
        report.append((r, k))
        worst = max(worst, r)
        if not f32:
            assert r <= BF16_VS_AUTOCAST * o['yard'][k] + BF16_FLOOR, (k, r, o['yard'][k])
    report.sort(reverse=True)
    med = report[len(report) // 2][0]
    print('\n%s %s: loss rel %.2e, gradient tensors: median %.2e, worst %s' % (
        kind, dtype, abs(float(loss) - o['loss']) / abs(o['loss']), med,
        ', '.join('%s %.2e%s' % (k.replace('decoder.', ''), r, '' if f32 else ' (autocast %.2e)' % o['yard'][k])
                  for r, k in report[:5])))
    assert worst <= (1e-3 if f32 else BF16_CEIL), report[:6]


def test_full_vocabulary_argmax_bit_exact_fp32():
    """adaptive_softmax.greedy (fused head + two tails + arg-max, never writes [N, 50265]) against the oracle's
    get_log_prob + argmax at the real vocabulary: ids identical, log-probs within 1e-5."""
    import tell_amd
    from oracle.functional import adaptive_log_probs
    from tell_amd.build import build_decoder
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(3)
    dec = build_decoder('flattened')
    asm = dec.adaptive_softmax
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 6, 1024, generator=g) * 2.0             # spread the logits: all three clusters win somewhere
    # make every cluster the winner for some rows
    with torch.no_grad():
        asm.head.class_proj.weight.mul_(6.0)
    tails = asm._tails()
    lp = adaptive_log_probs(x.reshape(-1, 1024), asm.cutoff, asm.head.word_proj.weight.detach(),
                            asm.head.class_proj.weight.detach(), [tails[0].detach(), tails[2].detach()],
                            [tails[1].detach(), tails[3].detach()])
    want_lp, want = lp.max(dim=-1)
    assert len({int(i >= 5000) + int(i >= 20000) for i in want.tolist()}) == 3, 'all clusters should win somewhere'
    dec.to(DEV)
    tok, got_lp = asm.greedy(x.to(DEV))
    assert torch.equal(tok.reshape(-1).cpu().long(), want)
    torch.testing.assert_close(got_lp.reshape(-1).cpu(), want_lp, rtol=1e-5, atol=1e-5)
    full = asm.get_log_prob(x.to(DEV)).reshape(-1, 50265).cpu()
    torch.testing.assert_close(full, lp, rtol=1e-4, atol=2e-5)
    # beam search's head: the 4 best of every row (register-resident top-k), best first == topk of the full row
    tk, tl = asm.topk(x.to(DEV), 4)
    want_tl, want_tk = lp.topk(4, dim=-1)
    assert torch.equal(tk.reshape(-1, 4).cpu().long(), want_tk)
    torch.testing.assert_close(tl.reshape(-1, 4).cpu(), want_tl, rtol=1e-5, atol=1e-5)


def test_generation_head_as_skinny_launches_matches_gemm_head_bf16():
    """The bf16 greedy head of a generation step (tell_amd/decode.py head_step: one skinny linear for head logits +
    cluster logits + the tails' projections, one per tail table, register-resident arg-max) against the GEMM head it
    replaces, at the real vocabulary, with every cluster winning somewhere: log-probs within bf16-accumulation-order
    noise, the same token wherever the two best log-probs are not a near tie."""
    import tell_amd
    from tell_amd import decode
    from tell_amd.build import build_decoder
    tell_amd.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(3)
    dec = build_decoder('flattened')
    asm = dec.adaptive_softmax
    with torch.no_grad():
        asm.head.class_proj.weight.mul_(6.0)
    dec.to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(32, 1, 1024, generator=g) * 2.0).to(DEV, torch.bfloat16)
    prev = decode.ENABLED
    try:
        decode.ENABLED = False
        tok0, lp0 = asm.greedy(x)
        decode.ENABLED = True
        tok1, lp1 = asm.greedy(x)
        tk1, tl1 = asm.topk(x, 4)                      # beam search's head: the 4 best of every row, best first
        decode.ENABLED = False
        tk0, tl0 = asm.topk(x, 4)
        decode.ENABLED = True
    finally:
        decode.ENABLED = prev
    torch.testing.assert_close(tl1.cpu(), tl0.cpu(), rtol=0, atol=2e-2)
    assert (tk1 == tk0).float().mean() >= 0.9 and bool((tl1[..., :-1] >= tl1[..., 1:]).all())
    assert torch.equal(tk1[..., 0].reshape(-1).cpu(), tok1.reshape(-1).cpu())          # top-1 == the greedy token
    tok0, tok1, lp0, lp1 = tok0.reshape(-1).cpu(), tok1.reshape(-1).cpu(), lp0.reshape(-1).cpu(), lp1.reshape(-1).cpu()
    assert len({int(i >= 5000) + int(i >= 20000) for i in tok1.tolist()}) == 3, 'all clusters should win somewhere'
    torch.testing.assert_close(lp1, lp0, rtol=0, atol=2e-2)
    same = tok0 == tok1
    assert same.float().mean() >= 0.9 and bool(((lp0 - lp1).abs()[~same] < 2e-2).all()), (tok0, tok1)


def test_full_size_greedy_decode_bit_exact_fp32_and_bf16_agreement():
    """The 4-context decoder at full size: (a) fp32 cached / graphed greedy ids == the oracle's reference-flow greedy
    ids, (b) the bf16 path under teacher forcing: fraction of positions whose arg-max equals the fp32 token, gated against a
    MEASURED yardstick - the oracle itself under torch.autocast(bfloat16) on the same tokens (random-init logits are nearly
    flat: what bf16 costs here is a property of the model, not a constant)."""
    import tell_amd
    from tell_amd.build import build_decoder
    o = _oracle('faces_objects')
    ref = o['ref'].eval()
    ctx, ids, _ = o['inputs']
    GEN = 10

    class _Shell:                                           # the generators live on the model class
        pass
    from oracle.models import CaptionModel as OModel
    from tell_amd.models.transformer import CaptionModel
    om = OModel.__new__(OModel)
    torch.nn.Module.__init__(om)
    om.decoder, om.padding_idx, om.index, om.sampling_topk, om.sampling_temp = ref, 1, 'roberta', 1, 1.0
    with torch.no_grad():
        _, want, _ = om._generate(ids[:, :1], {k: v.clone() for k, v in ctx.items()}, gen_len=GEN, eos=2)
    # the yardstick of the bf16 gates below: the SAME oracle under torch.autocast(bfloat16), teacher-forced on its own fp32
    # tokens and free-running - what bf16 arithmetic alone costs on this random-init model
    valid_w = (want[:, :-1] != 1) & (want[:, 1:] != 1)
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        y_out = ref({'roberta': want[:, :-1]}, {k: v.clone() for k, v in ctx.items()})
        y_tok = ref.get_normalized_probs((y_out[0], None), log_probs=True).argmax(-1)
        _, y_gen, _ = om._generate(ids[:, :1], {k: v.clone() for k, v in ctx.items()}, gen_len=GEN, eos=2)
    yard_tf = float((y_tok == want[:, 1:])[valid_w].float().mean())
    ny = min(y_gen.shape[1], want.shape[1])
    yard_free = float((y_gen[:, :ny] == want[:, :ny]).float().mean())
    one_pos = 1.0 / max(int(valid_w.sum()), 1)
    results = {}
    for dtype in (torch.float32, torch.bfloat16):
        tell_amd.set_compute_dtype(dtype)
        dec = build_decoder('faces_objects')
        dec.load_state_dict(o['sd'])
        dec.to(DEV).eval()
        m = CaptionModel.__new__(CaptionModel)
        torch.nn.Module.__init__(m)
        m.decoder, m.padding_idx, m.index, m.sampling_topk, m.sampling_temp = dec, 1, 'roberta', 1, 1.0
        m.training = False
        dctx = _to_dev(ctx, dtype)
        with torch.no_grad():
            if dtype == torch.float32:
                _, got, _ = m._generate_cached(ids[:, :1].to(DEV), dctx, gen_len=GEN, eos=2)
                assert torch.equal(got.cpu(), want[:, :got.shape[1]]), (got.cpu(), want)
            # teacher forcing on the oracle's tokens: position-wise arg-max of the full-sequence decoder
            seq = want[:, :-1].to(DEV)
            out = dec({'roberta': seq}, dctx)[0]
            tok, _ = dec.adaptive_softmax.greedy(out)
            valid = (want[:, :-1] != 1) & (want[:, 1:] != 1)
            results[dtype] = float((tok.cpu().long() == want[:, 1:])[valid].float().mean())
    # the bf16 generators themselves (captured decode step, device position counter): the weight-streaming step
    # (tell_amd/decode.py) and the layer-by-layer step, greedy and beam 4, against the fp32 oracle's greedy tokens
    from tell_amd import decode
    tell_amd.set_compute_dtype(torch.bfloat16)
    prev = decode.ENABLED
    gen = {}
    try:
        for fused in (True, False):
            decode.ENABLED = fused
            m.__dict__.pop('_decode_graphs', None)
            with torch.no_grad():
                for rep in range(2):                         # second call: every step is a graph replay
                    _, got, _ = m._generate_cached(ids[:, :1].to(DEV), dctx, gen_len=GEN, eos=2)
                _, beam, _ = m._generate_beam(ids[:, :1].to(DEV), dctx, 4, gen_len=GEN, eos=2)
            if tell_amd.graphs.ENABLED:
                hs = list(m.__dict__.get('_decode_graphs', {}).values())
                assert hs and all(h['graph'] not in (None, False) for h in hs), [h.get('error') for h in hs]
            gen[fused] = (got.cpu(), beam.cpu())
    finally:
        decode.ENABLED = prev
    n = min(gen[True][0].shape[1], want.shape[1])
    agree = {f: float((gen[f][0][:, :n] == want[:, :n]).float().mean()) for f in gen}
    print('\nbf16 captured greedy decode vs the fp32 oracle tokens: weight-streaming step %.3f, layer-by-layer %.3f'
          % (agree[True], agree[False]))
    print('autocast oracle: free-running %.3f, teacher-forced %.3f' % (yard_free, yard_tf))
    # (free-running: one flipped near-tie costs the rest of its row - at most one row of 4 more than the yardstick)
    assert agree[True] >= yard_free - 0.25 - 1e-6 and agree[True] >= agree[False] - 0.1, (agree, yard_free)
    nb = min(gen[True][1].shape[1], gen[False][1].shape[1])
    assert float((gen[True][1][:, :nb] == gen[False][1][:, :nb]).float().mean()) >= 0.8
    print('\nfull-size greedy, teacher-forced arg-max agreement with the fp32 oracle tokens: fp32 %.3f  bf16 %.3f'
          % (results[torch.float32], results[torch.bfloat16]))
    assert results[torch.float32] == 1.0
    assert results[torch.bfloat16] >= yard_tf - 2 * one_pos - 1e-6, (results, yard_tf)     # within two positions of the yardstick


def _inputs_batch(batch, seed):
    """4-context inputs at an arbitrary batch (ragged masks, some samples without faces)."""
    g = torch.Generator().manual_seed(seed)
    ctx = {}
    for n in ('image', 'article', 'faces', 'obj'):
        S, C = SHAPES[n]
        x = torch.randn(S, batch, C, generator=g) * 0.5
        if n in ('image', 'obj'):
            x = x.abs()
        lens = torch.randint(max(S // 2, 1), S + 1, (batch,), generator=g)
        if n == 'image':
            lens = torch.full((batch,), S)
        if n == 'faces':
            lens = torch.randint(0, S + 1, (batch,), generator=g)
            lens[0] = 0
        mask = torch.arange(S)[None, :] >= lens[:, None]
        ctx[n], ctx[n + '_mask'] = x * (~mask).t()[:, :, None], mask
    start = torch.zeros(batch, 1, dtype=torch.long)                       # <s>
    return ctx, start


def _shell_models(ref, dec):
    """The generators live on the model classes: bare CaptionModel shells around an oracle and a HIP decoder."""
    from oracle.models import CaptionModel as OModel
    from tell_amd.models.transformer import CaptionModel
    om = OModel.__new__(OModel)
    torch.nn.Module.__init__(om)
    om.decoder, om.padding_idx, om.index, om.sampling_topk, om.sampling_temp = ref, 1, 'roberta', 1, 1.0
    m = CaptionModel.__new__(CaptionModel)
    torch.nn.Module.__init__(m)
    m.decoder, m.padding_idx, m.index, m.sampling_topk, m.sampling_temp = dec, 1, 'roberta', 1, 1.0
    m.training = False
    return om, m


def _sharpened_eos(sd, factor):
    """A copy of the weights whose EOS row (tied input / output table, band 0) is scaled: its logit dominates wherever
    it is positive, so hypotheses end at different, data-dependent steps."""
    sd = {k: v.clone() for k, v in sd.items()}
    done = set()
    for k, v in sd.items():
        if (k.endswith('adaptive_softmax.head.word_proj.weight') or k.endswith('embed_tokens.embeddings.0.weight') or
                k.endswith('embedders.adaptive.embeddings.0.weight')) and v.data_ptr() not in done:
            v[2] *= factor
            v[1] = 0.0                                  # pad is never the arg-max (a trained model never emits it; its embedding row is zero in fairseq)
            done.add(v.data_ptr())
    return sd


def test_full_size_greedy_at_bench_batch_with_early_eos_bit_exact_fp32():
    """transformer_faces_objects.py:443-494 AT THE BENCH BATCH (B = 32): the cached / graphed fp32 greedy generator
    against the oracle's reference-flow loop, with an EOS row sharp enough that rows finish at different steps - the
    finished-row bookkeeping of tell_greedy_update (rows that have emitted </s> keep emitting pad, the loop ends when
    every row is done or at the cap) is compared at the batch the bench decodes, not only at fixture size."""
    import tell_amd
    from oracle.build import build_decoder as obuild
    from tell_amd.build import build_decoder
    BB, GEN = 32, 12
    o = _oracle('faces_objects')
    sd = _sharpened_eos(o['sd'], 12.0)          # (measured on the oracle: about one row in five ends within 12 steps)
    ref = obuild('faces_objects').eval()
    ref.load_state_dict({k: v for k, v in sd.items() if k in ref.state_dict()}, strict=False)
    ctx, start = _inputs_batch(BB, seed=41)
    tell_amd.set_compute_dtype(torch.float32)
    dec = build_decoder('faces_objects')
    dec.load_state_dict(sd)
    dec.to(DEV).eval()
    om, m = _shell_models(ref, dec)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        _, want, _ = om._generate(start, {k: v.clone() for k, v in ctx.items()}, gen_len=GEN, eos=2)
        _, got, _ = m._generate_cached(start.to(DEV), _to_dev(ctx, torch.float32), gen_len=GEN, eos=2)
    got = got.cpu()
    ended = (want == 2).any(1)
    first = torch.where(ended, (want == 2).float().argmax(1), torch.full((BB,), want.shape[1]))
    print('\nfull-size fp32 greedy at B=32: %d of %d rows end early (first </s> at steps %s), %d steps run'
          % (int(ended.sum()), BB, sorted(set(first[ended].tolist())), got.shape[1]))
    assert 3 <= int(ended.sum()) < BB and len(set(first.tolist())) >= 3, first    # genuinely ragged
    n = min(got.shape[1], want.shape[1])
    assert torch.equal(got[:, :n], want[:, :n]), (got, want)
    assert (got[:, n:] == 1).all() and (want[:, n:] == 1).all()
    for b in range(BB):                                                           # pad after the first </s>
        if ended[b]:
            assert (got[b, int(first[b]) + 1:] == 1).all()


def test_full_size_beam4_matches_oracle_definition_fp32():
    """BASELINE configs[4] is a beam-4 configuration: the cached beam generator (K/V shared by a sample's hypotheses,
    DynamicConv buffers reordered by parent - dynamic.py:338-342) AT FULL SIZE against the prefix-re-decoding
    definition of oracle/beam.py, fp32: identical token ids and summed log-probabilities; also beam 2."""
    import tell_amd
    from oracle.beam import beam_search
    from oracle.build import build_decoder as obuild
    from tell_amd.build import build_decoder
    BB, GEN = 2, 8
    o = _oracle('faces_objects')
    sd = _sharpened_eos(o['sd'], 2.0)
    ref = obuild('faces_objects').eval()
    ref.load_state_dict({k: v for k, v in sd.items() if k in ref.state_dict()}, strict=False)
    ctx, start = _inputs_batch(BB, seed=43)
    tell_amd.set_compute_dtype(torch.float32)
    dec = build_decoder('faces_objects')
    dec.load_state_dict(sd)
    dec.to(DEV).eval()
    om, m = _shell_models(ref, dec)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        for K in (4, 2):
            ref_ids, ref_score = beam_search(om, start, {k: v.clone() for k, v in ctx.items()}, K, gen_len=GEN)
            lp, got, _ = m._generate_beam(start.to(DEV), _to_dev(ctx, torch.float32), K, gen_len=GEN, eos=2)
            got = got.cpu()
            n = min(got.shape[1], ref_ids.shape[1])
            assert torch.equal(got[:, :n], ref_ids[:, :n]), (K, got, ref_ids)
            assert (got[:, n:] == 1).all() and (ref_ids[:, n:] == 1).all()
            assert torch.allclose(lp.sum(1).cpu(), ref_score, rtol=1e-4, atol=5e-4), (lp.sum(1), ref_score)


@pytest.mark.parametrize('kind,beams', [('faces_objects', 1), ('faces_objects', 4), ('faces_objects', 16), ('faces_objects', 56),
                                        ('flattened', 4)])
def test_fused_decode_step_matches_layer_by_layer_step_and_fp32(kind, beams):
    """The generation step as weight-streaming launches (tell_amd/decode.py, csrc/decode.hip: skinny linears with
    LayerNorm prologues, DynamicConv step, grouped one-query attention) against (a) the layer-by-layer bf16 step it
    replaces and (b) the fp32 full-sequence decoder (itself bit-exact-greedy against the oracle above), teacher-forced
    on the oracle's tokens.  beams > 1: the hypotheses of a sample (rows b*beams + j) share its K/V cache; beams = 16 / 56:
    64 / 224 rows = two / seven 32-row groups per column tile (the step takes up to decode.MAX_ROWS = 256 rows)."""
    import tell_amd
    from tell_amd import decode
    from tell_amd.build import build_decoder
    o = _oracle(kind)
    ctx, ids, _ = o['inputs']
    STEPS = 8
    seq = ids[:, :STEPS].to(DEV)
    tell_amd.set_compute_dtype(torch.float32)
    dec32 = build_decoder(kind)
    dec32.load_state_dict(o['sd'])
    dec32.to(DEV).eval()
    with torch.no_grad():
        want = dec32({'roberta': seq}, _to_dev(ctx, torch.float32))[0].float()            # [B, STEPS, E]
    del dec32
    tell_amd.set_compute_dtype(torch.bfloat16)
    dec = build_decoder(kind)
    dec.load_state_dict(o['sd'])
    dec.to(DEV).eval()
    dctx = _to_dev(ctx, torch.bfloat16)
    outs = {}
    prev, prev_rows = decode.ENABLED, decode.MAX_ROWS
    decode.MAX_ROWS = 256
    try:
        with torch.no_grad():
            kv = dec.project_contexts(dctx)
            for fused in (False, True):
                decode.ENABLED = fused
                state = dec.static_incremental_state(B * beams, DEV, torch.bfloat16)
                xs = []
                for i in range(STEPS):
                    cur = seq[:, i:i + 1].repeat_interleave(beams, dim=0).contiguous()
                    assert decode.usable(dec, torch.empty(1, B * beams, 1024, dtype=torch.bfloat16, device=DEV),
                                         state, kv) == fused
                    xs.append(dec({'roberta': cur}, dctx, incremental_state=state, kv_cache=kv)[0][:, 0].float())
                outs[fused] = torch.stack(xs, 1)                                           # [B*beams, STEPS, E]
    finally:
        decode.ENABLED, decode.MAX_ROWS = prev, prev_rows
    w = want.repeat_interleave(beams, dim=0)
    e_fused, e_plain, e_pair = _rel(outs[True], w), _rel(outs[False], w), _rel(outs[True], outs[False])
    print('\ndecode step, beams %d: fused vs fp32 %.3e, layer-by-layer vs fp32 %.3e, fused vs layer-by-layer %.3e'
          % (beams, e_fused, e_plain, e_pair))
    assert e_fused < BF16_OUT and e_fused < 1.25 * e_plain + 1e-3, (e_fused, e_plain)
    if beams > 1:                                   # identical tokens, identical state: the rows of a sample agree
        r = outs[True].view(B, beams, STEPS, -1)
        assert torch.equal(r, r[:, :1].expand_as(r))


def test_layer_by_layer_decode_step_above_128_rows_matches_fp32():
    """More than 128 rows (batch x beam) take the layer-by-layer step (decode.MAX_ROWS): the training GEMM kernels for the
    linears, and - since round 5 - the decode kernels for the DynamicConv step (tell_dynconv_step) and the n context
    attentions (one tell_attn_decode launch).  192 rows = 4 samples x 48 hypotheses against the fp32 full-sequence decoder,
    teacher-forced; the hypotheses of a sample see identical tokens and must agree bit for bit."""
    import tell_amd
    from tell_amd import decode
    from tell_amd.build import build_decoder
    kind, beams, STEPS = 'faces_objects', 48, 6
    prev_rows, decode.MAX_ROWS = decode.MAX_ROWS, 128          # (the default is 256 since round 6: pin the layer path for 192 rows)
    o = _oracle(kind)
    ctx, ids, _ = o['inputs']
    seq = ids[:, :STEPS].to(DEV)
    tell_amd.set_compute_dtype(torch.float32)
    dec32 = build_decoder(kind)
    dec32.load_state_dict(o['sd'])
    dec32.to(DEV).eval()
    with torch.no_grad():
        want = dec32({'roberta': seq}, _to_dev(ctx, torch.float32))[0].float()            # [B, STEPS, E]
    del dec32
    tell_amd.set_compute_dtype(torch.bfloat16)
    dec = build_decoder(kind)
    dec.load_state_dict(o['sd'])
    dec.to(DEV).eval()
    dctx = _to_dev(ctx, torch.bfloat16)
    names = []
    with torch.no_grad():
        kv = dec.project_contexts(dctx)
        state = dec.static_incremental_state(B * beams, DEV, torch.bfloat16)
        assert not decode.usable(dec, torch.empty(1, B * beams, 1024, dtype=torch.bfloat16, device=DEV), state, kv)
        xs = []
        for i in range(STEPS):
            cur = seq[:, i:i + 1].repeat_interleave(beams, dim=0).contiguous()
            if i == 1:                                      # which kernels a step launches (torch.profiler / roctracer)
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as pf:
                    xs.append(dec({'roberta': cur}, dctx, incremental_state=state, kv_cache=kv)[0][:, 0].float())
                    torch.cuda.synchronize()
                names = [e.name for e in pf.events()]
            else:
                xs.append(dec({'roberta': cur}, dctx, incremental_state=state, kv_cache=kv)[0][:, 0].float())
        out = torch.stack(xs, 1)                                                           # [B*beams, STEPS, E]
    w = want.repeat_interleave(beams, dim=0)
    err = _rel(out, w)
    print('\ndecode step at %d rows (layer by layer + decode kernels) vs fp32: %.3e' % (B * beams, err))
    assert err < BF16_OUT, err
    r = out.view(B, beams, STEPS, -1)
    assert torch.equal(r, r[:, :1].expand_as(r))
    decode.MAX_ROWS = prev_rows
    joined = ' '.join(names)
    assert 'dynconv_step_kernel' in joined and 'attn_decode_kernel' in joined, joined


# --------------------------------------------------------------------------------------------------------------------
# The shapes bench.py times, through the captured step graph.
# B = 4 above runs the per-token GEMMs at M = 128 rows (register-staged 64x64 kernel); at B = 32 / 16 they have
# M = 1024 / 512 rows: the direct-to-LDS 64x64 / 128x128 kernels, the grouped query / output projections, the grouped
# weight-gradient launches with 1024-row reductions, the 256x256 ping-pong kernel for the article K|V projection (B = 32).
# --------------------------------------------------------------------------------------------------------------------
class _StandInRoberta(torch.nn.Module):
    """Three 'hidden states' looked up from a table: the article encoder as far as the decoder half is concerned."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.register_buffer('tab', torch.randn(3, 64, 1024, generator=g) * 0.5)

    def extract_features(self, ids, return_all_hiddens=False):
        import tell_amd
        out = self.tab[:, ids % 64]
        if out.is_cuda:
            return out.to(tell_amd.compute_dtype())
        return list(out)


class _StandInResnet(torch.nn.Module):
    def __init__(self, nhwc):
        super().__init__()
        g = torch.Generator().manual_seed(4)
        self.register_buffer('proj', torch.randn(2048, 3, generator=g) * 0.3)
        self.nhwc = nhwc

    def forward(self, image):
        import tell_amd
        f = torch.relu(torch.einsum('oc,bchw->bohw', self.proj, torch.nn.functional.avg_pool2d(image, 32)))
        if not self.nhwc:
            return f
        return f.permute(0, 2, 3, 1).reshape(f.shape[0], 49, 2048).to(tell_amd.compute_dtype()).contiguous()


@pytest.mark.parametrize('kind,batch,dyn', [('faces_objects', 32, False), ('flattened', 16, False), ('faces_objects', 32, True)],
                         ids=['faces_objects-32', 'flattened-16', 'faces_objects-32-tile-queue'])
def test_bench_shape_step_through_the_step_graph_matches_oracle(kind, batch, dyn, monkeypatch):
    """BASELINE configs[2] (4 contexts, B = 32) and configs[1] (2 contexts, B = 16) at full decoder size, 512-token
    articles, 33-token captions, bf16: one optimisation step's decoder half - loss and EVERY gradient tensor (decoder
    and the RoBERTa layer-mix weights) - as the captured step graph REPLAYS it (training/step_graph.py), against the
    CPU oracle on the same batch; reference: decoder_faces_objects.py:255-365, transformer_faces_objects.py:67-140.
    Bounds as in the B = 4 test: each gradient within BF16_VS_AUTOCAST x the error the oracle itself shows under CPU
    bf16 autocast on this batch.  (The encoders are table look-ups here; their own bench-size parity lives in
    test_gpu_encoders.py.)
    tile-queue: the same step with TELL_Q4_DYNAMIC=1 - per-XCD tile counters in the resident GEMMs (the article K|V
    projection here), the DEFAULT under data parallelism (training/trainer.py) - captured into the graph: its slots come
    out of the per-device queue (csrc/gemm.hip) and go back when the graph entry is dropped."""
    import tell_amd
    from oracle.build import build_model as obuild
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    fo = kind == 'faces_objects'
    if dyn:
        tell_amd.hip.set_option('q4_dynamic', 1)
    held0 = tell_amd.hip.tile_queue_stats() if torch.cuda.is_available() else None
    tell_amd.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    gpu = build_model(kind, _StandInResnet(True), _StandInRoberta(), n_bert_layers=3)
    _no_dropout(gpu)
    cpu = obuild(kind, _StandInResnet(False), _StandInRoberta(), n_bert_layers=3).train()
    cpu.load_state_dict({k: v for k, v in gpu.state_dict().items() if k in cpu.state_dict()}, strict=False)
    _no_dropout(cpu)
    bt = synthetic_batch(B=batch, article_len=512, caption_len=33, faces_objects=fo, seed=77, variable=True)
    clone = lambda x: {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in x.items()}      # noqa: E731
    # ---- oracle: fp32, then the bf16-autocast yardstick
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = cpu(**clone(bt))
    ref['loss'].backward()
    want = {k: p.grad.detach().clone() for k, p in cpu.named_parameters() if p.grad is not None}
    cpu.zero_grad(set_to_none=True)
    with torch.autocast('cpu', dtype=torch.bfloat16):
        ref16 = cpu(**clone(bt))
    ref16['loss'].float().backward()
    yard = {k: _rel(p.grad, want[k]) for k, p in cpu.named_parameters()
            if p.grad is not None and want[k].norm().item() >= 1e-12}
    # ---- HIP: eager step (records the graph), then the REPLAY whose gradients are compared
    tr = Trainer(gpu, dict(lr=0.0, warmup=-1, t_total=-1), device=DEV, capture_after=1)
    tr.defer_update = True
    dev = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV)) for k, v in bt.items()}
    l0 = float(tr.train_one_batch(clone(dev)))
    tr.flat.zero_grad()
    loss = float(tr.train_one_batch(clone(dev)))
    torch.cuda.synchronize()
    e = next(iter(tr.step_graph.entries.values()))
    assert e['state'] == 'ready' and tr.step_graph.replays == 1, e.get('error')
    assert abs(loss - l0) <= 1e-5 * abs(l0), (loss, l0)                    # eager and replay run the same kernels
    assert abs(loss - float(ref['loss'])) <= BF16_LOSS * abs(float(ref['loss'])), (loss, float(ref['loss']))
    assert int(e['sample_size']) == int(ref['sample_size'])
    pd = dict(gpu.named_parameters())
    report, worst = [], 0.0
    for k, gref in want.items():
        g = pd[k].grad
        assert g is not None, k
        if gref.norm().item() < 1e-12:
            assert g.float().abs().max().item() <= 1e-6, k
            continue
        r = _rel(g, gref)
        report.append((r, k))
        worst = max(worst, r)
        assert r <= BF16_VS_AUTOCAST * yard[k] + BF16_FLOOR, (k, r, yard[k])
    report.sort(reverse=True)
    print('\n%s B=%d through the step graph: loss rel %.2e; %d gradient tensors, median %.2e, worst %s' % (
        kind, batch, abs(loss - float(ref['loss'])) / abs(float(ref['loss'])), len(report), report[len(report) // 2][0],
        ', '.join('%s %.2e (autocast %.2e)' % (k.replace('decoder.', ''), r, yard[k]) for r, k in report[:4])))
    assert worst <= BF16_CEIL, report[:6]
    if dyn:
        slots = list(e['tile_slots'].slots)
        del e
        assert slots, 'the captured step took no tile-counter slot: the dynamic path did not run'
        n, fresh, free = tell_amd.hip.tile_queue_stats()
        assert fresh - free >= len(slots) and n == (1 << 20) // 8
        tr.step_graph.reset()                                              # dropping the graph gives its slots back
        import gc
        gc.collect()
        n2, fresh2, free2 = tell_amd.hip.tile_queue_stats()
        assert fresh2 == fresh and free2 == free + len(slots), (held0, (n, fresh, free), (n2, fresh2, free2))
