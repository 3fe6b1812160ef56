"""GPU: the data plane feeding the model (shards -> reader -> bucket iterator -> device batches -> training steps) and
the evaluation tail (evaluate loop -> generations.jsonl + BLEU bookkeeping, tell/commands/evaluate.py:89-223)."""
import json
import os

import numpy as np
import pytest
import torch

from test_gpu_train import DEV, KW, _Res, _Rob, _no_dropout

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu():
    import tell_amd
    tell_amd.hip.require_gpu()
    yield
    torch.cuda.synchronize()


def _shards(tmp_path, n=24, vocab=600, seed=0):
    from tell_amd.data import write_shard
    g = np.random.RandomState(seed)
    for split in ('train', 'test'):
        samples = []
        for i in range(n):
            L, T, F, O = g.randint(8, 24), g.randint(5, 12), g.randint(0, 5), g.randint(0, 7)
            samples.append({
                'context_ids': np.r_[0, g.randint(4, vocab, L - 2), 2], 'caption_ids': np.r_[0, g.randint(4, vocab, T - 2), 2],
                'image': g.randint(0, 256, (224, 224, 3)).astype(np.uint8),
                'face_embeds': g.randn(F, 512).astype(np.float32), 'obj_embeds': np.abs(g.randn(O, 2048)).astype(np.float32),
                'metadata': {'caption': 'word%d word%d again' % (i, i + 1), 'context': 'ctx', 'web_url': 'u%d' % i,
                             'image_path': '%d.jpg' % i, 'image_pos': 0}})
        write_shard(str(tmp_path / ('%s-00000.npz' % split)), samples)


def test_image_normalize_kernel_matches_host_path():
    from tell_amd.data.iterators import normalize_images
    img = np.random.RandomState(1).randint(0, 256, (3, 224, 224, 3)).astype(np.uint8)
    torch.testing.assert_close(normalize_images(img, DEV).cpu(), normalize_images(img, 'cpu'), rtol=1e-6, atol=1e-6)


def test_shards_to_training_steps_and_evaluation_tail(tmp_path):
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.commands import evaluate
    from tell_amd.data import BucketIterator, DatasetReader
    from tell_amd.training import Trainer
    _shards(tmp_path)
    tell_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    model = build_model('faces_objects', _Res(True), _Rob(1024), n_bert_layers=3, **KW)
    _no_dropout(model)
    reader = DatasetReader.by_name('nytimes_faces_ner_matched')(use_objects=True, shard_dir=str(tmp_path))
    it = BucketIterator(sorting_keys=[['context', 'num_tokens'], ['caption', 'num_tokens']], batch_size=8,
                        maximum_samples_per_batch=['num_tokens', 16384])
    tr = Trainer(model, dict(lr=5e-3, warmup=-1, t_total=-1, max_grad_norm=1.0), device=DEV)
    losses = []
    for epoch in range(3):
        for batch in it(reader._read('train'), shuffle=True, device=DEV):
            n = batch['context']['roberta'].shape[0]
            assert batch['image'].is_cuda and batch['face_embeds'].shape[0] == n and len(batch['metadata']) == n
            losses.append(float(tr.train_one_batch(batch)))
    assert all(l == l for l in losses) and sum(losses[-3:]) < sum(losses[:3])
    # ---- evaluation tail
    model.get_metrics(reset=True)                 # (the trainer resets the running counters at every epoch end)
    out_dir = str(tmp_path / 'serialization')
    metrics = evaluate(model, reader._read('test'), BucketIterator(sorting_keys=[['context', 'num_tokens']], batch_size=8),
                       DEV, out_dir, eval_suffix='_t')
    assert metrics['_n_samples'] == 24 and metrics['loss'] == metrics['loss']
    assert all(('bleu-%d' % k) in metrics for k in (1, 2, 3, 4))
    lines = [json.loads(ln) for ln in open(os.path.join(out_dir, 'generations_t.jsonl'))]
    assert len(lines) == 24
    for rec in lines:
        assert {'caption', 'raw_caption', 'generation', 'copied_texts', 'web_url', 'image_path', 'context',
                'caption_np', 'gen_np'} <= set(rec)
        assert rec['raw_caption'].startswith('word') and isinstance(rec['generation'], str)
    assert model.evaluate_mode is True
    with pytest.raises(AssertionError):            # the reference refuses to append to an existing generations file
        evaluate(model, reader._read('test'), it, DEV, out_dir, eval_suffix='_t')


@pytest.mark.parametrize('beam', [1, 3])
def test_generate_stream_equals_batch_by_batch_generation(beam):
    """The pipelined test-set loop (encoders of batch N+1 on their own streams underneath the decode loop of batch N)
    yields, batch for batch, the ids and log-probabilities of `generate` called on that batch alone - bit for bit: the
    encoders read nothing the decode loop writes, and a batch's encoder outputs live in their own buffer set."""
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    tell_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        model = build_model('faces_objects', _Res(True), _Rob(1024), n_bert_layers=3, **KW).to(DEV).eval()
        batches = [synthetic_batch(4, 24 + 8 * (i % 2), 9, True, seed=11 + i, device=DEV, vocab=600, cutoffs=(100, 300)) for i in range(5)]

        def clone(b):
            return {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}
        alone = [model.generate(**clone(b), beam_size=beam) for b in batches]
        torch.cuda.synchronize()
        seen = 0
        for i, (b, out) in enumerate(model.generate_stream((clone(b) for b in batches), beam_size=beam)):
            assert torch.equal(out['gen_ids'], alone[i]['gen_ids'])
            assert torch.equal(out['log_probs'], alone[i]['log_probs'])
            seen += 1
        assert seen == len(batches)
    finally:
        tell_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize('beam,lanes', [(1, 2), (4, 2), (1, 3)])
def test_generate_lanes_equals_batch_by_batch_generation(beam, lanes):
    """Several decode loops in flight together (CaptionModel.generate_lanes: one stream, one captured step, one set of static
    buffers, counters and split-reduction workspace per lane; the host alternates the lanes' graph replays) yield, batch for
    batch, the ids and log-probabilities of `generate` called on that batch alone - bit for bit, twice over (the second
    pass replays every lane's captured step from its first token on), with an odd number of batches (a last group with
    fewer batches than lanes) and two context shapes."""
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    tell_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        model = build_model('faces_objects', _Res(True), _Rob(1024), n_bert_layers=3, **KW).to(DEV).eval()
        batches = [synthetic_batch(4, 24 + 8 * (i % 2), 9, True, seed=31 + i, device=DEV, vocab=600, cutoffs=(100, 300)) for i in range(5)]

        def clone(b):
            return {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}
        alone = [model.generate(**clone(b), beam_size=beam) for b in batches]
        torch.cuda.synchronize()
        for rep in range(2):
            seen = 0
            for i, (b, out) in enumerate(model.generate_lanes((clone(b) for b in batches), beam_size=beam, lanes=lanes)):
                torch.cuda.synchronize()
                assert torch.equal(out['gen_ids'], alone[i]['gen_ids']), (rep, i)
                assert torch.equal(out['log_probs'], alone[i]['log_probs']), (rep, i)
                seen += 1
            assert seen == len(batches)
    finally:
        tell_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize('beam', [1, 3])
def test_evaluate_forward_through_lanes_equals_forward_batch_by_batch(beam):
    """What commands/evaluate.py drives: `generate_lanes(forward=True)` yields, batch for batch, the evaluate-mode output of
    `forward` (the loss, the generated ids, the detokenised texts, the captions) and leaves the model's running metrics
    (sample count, batch count, per-sample BLEU sums) exactly where batch-by-batch `forward` calls leave them."""
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    tell_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        model = build_model('faces_objects', _Res(True), _Rob(1024), n_bert_layers=3, **KW).to(DEV).eval()
        model.evaluate_mode = True
        model.eval_beam_size = beam
        batches = [synthetic_batch(4, 24 + 8 * (i % 2), 9, True, seed=51 + i, device=DEV, vocab=600, cutoffs=(100, 300)) for i in range(3)]
        for i, b in enumerate(batches):
            b['metadata'] = [{'caption': '7 8 9 %d 11' % (10 + j + i)} for j in range(4)]

        def clone(b):
            return {k: (dict(v) if isinstance(v, dict) else (list(v) if isinstance(v, list) else v.clone())) for k, v in b.items()}
        with torch.no_grad():
            alone = [model(**clone(b)) for b in batches]
        torch.cuda.synchronize()
        want = (model.n_samples, model.n_batches, dict(model.sample_history))
        model.get_metrics(reset=True)
        model.n_samples = model.n_batches = 0
        seen = 0
        for i, (b, out) in enumerate(model.generate_lanes((clone(b) for b in batches), lanes=2, forward=True)):
            assert torch.equal(out['loss'], alone[i]['loss']), i
            assert (out['gen_ids'] == alone[i]['gen_ids']).all() and out['gen_ids'].shape == alone[i]['gen_ids'].shape, i
            assert out['generations'] == alone[i]['generations'] and out['captions'] == alone[i]['captions'], i
            seen += 1
        assert seen == len(batches)
        assert (model.n_samples, model.n_batches, dict(model.sample_history)) == want
    finally:
        tell_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize('beam', [1, 4])
def test_several_decode_steps_per_graph_replay_equal_one_step_per_replay(beam):
    """From its first all-finished test on, a generation loop replays ONE graph that holds `check_every` consecutive decode
    steps (CaptionModel._decode_stepper: step.multi - the bookkeeping launch that ends a recorded step leaves the next
    step's position offset in the device counter, so recorded steps chain on the device as single replays do).  Ids and
    log-probabilities must equal the one-step-per-replay loop bit for bit, over two generations of the same stepper (the
    second replays every graph from its first token on), and the multi-step graph must really have been recorded."""
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.models import transformer as tr
    tell_amd.set_compute_dtype(torch.bfloat16)
    keep = tr.MULTI_STEP_GRAPHS
    try:
        torch.manual_seed(0)
        model = build_model('faces_objects').to(DEV).eval()       # (full size: the step with in-graph bookkeeping; random
        batches = [synthetic_batch(4, 64, 9, True, seed=71 + i, device=DEV) for i in range(2)]   # weights never emit </s>: 100 steps)

        def clone(b):
            return {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}
        tr.MULTI_STEP_GRAPHS = False
        single = [model.generate(**clone(b), beam_size=beam) for b in batches]
        torch.cuda.synchronize()
        model.__dict__['_decode_graphs'].clear()
        tr.MULTI_STEP_GRAPHS = True
        for rep in range(2):
            for i, b in enumerate(batches):
                out = model.generate(**clone(b), beam_size=beam)
                torch.cuda.synchronize()
                assert torch.equal(out['gen_ids'], single[i]['gen_ids']), (rep, i)
                assert torch.equal(out['log_probs'], single[i]['log_probs']), (rep, i)
        recorded = [k for h in model.__dict__['_decode_graphs'].values() for k in h if isinstance(k, tuple) and k[:1] == ('multi',) and len(k) == 2]
        assert recorded and all(h[k] for h in model.__dict__['_decode_graphs'].values() for k in recorded if k in h), \
            [h.get('multi_error') for h in model.__dict__['_decode_graphs'].values()]
    finally:
        tr.MULTI_STEP_GRAPHS = keep
        tell_amd.set_compute_dtype(torch.float32)
