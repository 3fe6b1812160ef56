"""tell/models/transformer_faces_objects.py:23-517 and tell/models/transformer_flattened.py:24-443
on the MI355X path (training forward + loss, greedy generation)."""
import math
import os
from collections import defaultdict

import torch
import torch.nn as nn

from .. import graphs, ops, streams
from ..common.registrable import Registrable

# decode steps per graph replay once a generation loop is past its first all-finished test (CaptionModel._decode_stepper: multi)
MULTI_STEP_GRAPHS = os.environ.get('TELL_MULTI_STEP_GRAPHS', '1') != '0'


_OVERLAP = os.environ.get('TELL_ENCODER_OVERLAP', '1') != '0'
# where the PREFETCHED ResNet pass of the next batch is enqueued: 'own' = its side stream (three streams share the chip),
# 'main' = the training stream, in front of the decoder step of the current batch (two streams: RoBERTa against ResNet +
# decoder back to back - 4.3 + 6.7 ms against 11.2 ms alone).  MEASURED in round 6: tools/step_timeline.py, DESIGN.md.
_RESNET_STREAM = os.environ.get('TELL_RESNET_STREAM', 'own')
def _side_stream(device, name='resnet'):
    return streams.get(name, device)


class EncodedBatch:
    """Outputs of the frozen encoders for one batch (CaptionModel.encode), possibly still in flight on the
    encoder streams."""

    def __init__(self):
        self.stack = self.x_image = self.article_mask = None
        self.events = []
        self.static = False      # stack / x_image are graph-owned static buffers (stable addresses across steps)
        self.slots = []          # (graph slot, generation at production): a later replay of the slot overwrites the data

    def stale(self):
        """True when a graph replay issued after this batch was encoded has reused one of its output buffers."""
        return any(s.get('generation') != g for s, g in self.slots)

    def wait(self):
        """Join the producing streams into the current stream (idempotent)."""
        if self.events:
            cur = torch.cuda.current_stream()
            for ev in self.events:
                cur.wait_event(ev)
            self.events = []
            for t in (self.x_image, self.article_mask, self.stack):
                for u in (t if isinstance(t, (list, tuple)) else (t,)):
                    if torch.is_tensor(u) and u.is_cuda:
                        u.record_stream(cur)


class Model(nn.Module, Registrable):
    """Stand-in for allennlp.models.Model (forward(**batch) -> dict with 'loss')."""

    def __init__(self, vocab=None):
        super().__init__()
        self.vocab = vocab

    def get_metrics(self, reset=False):
        return {}

    def decode(self, output_dict):
        return output_dict


class CaptionModel(Model):
    USE_FACES_OBJECTS = False
    EXTRA_CONTEXTS = ()          # which of ('faces', 'obj') the model feeds to its decoder

    def __init__(self, vocab, decoder, criterion, evaluate_mode=False, attention_dim=1024, hidden_size=1024,
                 dropout=0.1, vocab_size=50264, model_name='roberta-base', namespace='bpe', index='roberta',
                 padding_value=1, use_context=True, sampling_topk=1, sampling_temp=1.0, weigh_bert=False,
                 initializer=None, resnet=None, roberta=None, n_bert_layers=25):
        super().__init__(vocab)
        self.decoder, self.criterion = decoder, criterion
        self.index, self.namespace = index, namespace
        if resnet is None:
            from .resnet import resnet152
            resnet = resnet152()
        if roberta is None:
            from .roberta import roberta_large
            roberta = roberta_large()
        self.resnet, self.roberta = resnet, roberta
        self.use_context = use_context
        self.padding_idx = padding_value
        self.evaluate_mode = evaluate_mode
        self.sampling_topk, self.sampling_temp = sampling_topk, sampling_temp
        if sampling_topk != 1:
            raise NotImplementedError('generation is greedy (sampling_topk: 1 in every config)')
        self.weigh_bert = weigh_bert
        if weigh_bert:
            self.bert_weight = nn.Parameter(torch.rand(n_bert_layers))      # nn.init.uniform_, :57-59
        self.n_batches = 0
        self.n_samples = 0
        self.sample_history = defaultdict(float)
        # captured encoder / decode graphs bake in the addresses of working copies of the weights: anything that can
        # re-home those copies (a checkpoint load, a trainer re-flagging requires_grad) drops the captures
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.reset_graphs())

    def reset_graphs(self):
        """Forget every captured hipGraph of this model (encoders, decode steps); they are re-recorded on next use."""
        for k in ('_resnet_graph', '_roberta_graph'):
            g = self.__dict__.get(k)
            if g is not None:
                g.reset()
        self.__dict__.pop('_decode_graphs', None)
        self.__dict__.pop('_decode_graphs_stamp', None)

    def set_capture_after(self, n):
        """Sightings of an input shape before the encoder graphs capture it (graphs.CAPTURE_AFTER by default)."""
        self.__dict__['_capture_after'] = n
        self.reset_graphs()
        for k in ('_resnet_graph', '_roberta_graph'):
            self.__dict__.pop(k, None)

    def _run_resnet(self, image, slot=None):
        """The frozen trunk as one hipGraph replay per step (graphs.GraphedCall); eager for the first call."""
        from .resnet import ResNetFeatureExtractor
        if not isinstance(self.resnet, ResNetFeatureExtractor):      # a user-supplied trunk: no assumptions
            return self.resnet(image)
        g = self.__dict__.get('_resnet_graph')
        if g is None:
            g = self.__dict__['_resnet_graph'] = graphs.GraphedCall(self.resnet, 'resnet152',
                                                                    capture_after=self.__dict__.get('_capture_after'))
        w = self.resnet.conv1.weight                 # a reloaded / moved / re-typed trunk must not replay stale pointers
        from .resnet import stats_epoch
        # (eval captures bake in weights folded with the running statistics: a train-mode pass in between retires them)
        return g(image, key=(self.resnet.training, ops.rt.compute_dtype(), w._version, w.data_ptr(),
                             0 if self.resnet.training else stats_epoch()), slot=slot)

    def _run_roberta(self, article_ids, slot=None):
        """RoBERTa-large (~170 launches, dropout active in train mode) as one hipGraph replay per step."""
        from .roberta import RobertaEncoder
        if not isinstance(self.roberta, RobertaEncoder):
            return self.roberta.extract_features(article_ids, return_all_hiddens=True)
        g = self.__dict__.get('_roberta_graph')
        if g is None:
            g = self.__dict__['_roberta_graph'] = graphs.GraphedCall(
                lambda ids: self.roberta.extract_features(ids, return_all_hiddens=True), 'roberta-large', rng=True,
                capture_after=self.__dict__.get('_capture_after'))
        w = self.roberta.model.decoder.sentence_encoder.layers[0].fc1.weight
        return g(article_ids, key=(self.roberta.training, ops.rt.compute_dtype(), w._version, w.data_ptr()), slot=slot)

    # ---- frozen encoders -------------------------------------------------------------
    def encode(self, context, image, ahead=False):
        """ResNet-152 + RoBERTa-large on one batch (transformer_faces_objects.py:335-353) -> EncodedBatch.

        The two encoders are independent and read no trainable weight.  ahead=False: RoBERTa runs on the current
        stream, ResNet's many small launches on a side stream underneath RoBERTa's chip-filling GEMMs
        (TELL_ENCODER_OVERLAP=0 serialises them).  ahead=True: both run on their own streams and the call returns
        at once - the trainer uses it to encode batch N+1 underneath the latency-bound decoder forward /
        backward / optimizer of batch N; `EncodedBatch.wait()` joins them into the consumer's stream."""
        with torch.no_grad():
            main = torch.cuda.current_stream()
            article_ids = context[self.index]
            enc = EncodedBatch()
            # both encoder graphs replay into the buffer set of this call's parity: consecutive batches never share a
            # buffer, and the addresses the decoder step graph is keyed on depend on the parity alone (graphs.py)
            par = self.__dict__['_enc_parity'] = self.__dict__.get('_enc_parity', -1) + 1
            if ahead:
                rs = _side_stream(image.device, 'roberta')
                is_ = main if _RESNET_STREAM == 'main' else _side_stream(image.device, 'resnet')
                start = torch.cuda.Event()
                start.record(main)
                rs.wait_event(start)
                if is_ is not main:
                    is_.wait_event(start)
                with torch.cuda.stream(rs), ops.hip.bound_stream():
                    enc.article_mask = self._pad_mask(article_ids)                      # :347
                    enc.stack = self._run_roberta(article_ids, par)
                    article_ids.record_stream(rs)
                    enc.events.append(torch.cuda.Event())
                    enc.events[-1].record(rs)
                with torch.cuda.stream(is_), ops.hip.bound_stream():
                    enc.x_image = self._run_resnet(image, par)
                    image.record_stream(is_)
                    enc.events.append(torch.cuda.Event())
                    enc.events[-1].record(is_)
                enc.static = self._encoders_replayed()
                self._note_slots(enc)
                return enc
            side = _side_stream(image.device, 'resnet') if _OVERLAP else None
            enc.article_mask = self._pad_mask(article_ids)                              # :347
            if side is not None:
                start = torch.cuda.Event()
                start.record(main)
            # RoBERTa is issued FIRST: its ~300 launches keep the main stream busy for ~10 ms of GPU time
            # while the host is still issuing ResNet's small launches onto the side stream.
            enc.stack = self._run_roberta(article_ids, par)                                   # [L,B,S,E]
            if side is not None:
                side.wait_event(start)
                with torch.cuda.stream(side), ops.hip.bound_stream():
                    enc.x_image = self._run_resnet(image, par)             # [B,49,2048] (NHWC == :335-341)
                    enc.events.append(torch.cuda.Event())
                    enc.events[-1].record(side)
            else:
                enc.x_image = self._run_resnet(image, par)
            enc.static = self._encoders_replayed()
            self._note_slots(enc)
            return enc

    def _encoders_replayed(self):
        return all(getattr(self.__dict__.get(k), 'last_replayed', False) for k in ('_resnet_graph', '_roberta_graph'))

    def _note_slots(self, enc):
        for k in ('_resnet_graph', '_roberta_graph'):
            g = self.__dict__.get(k)
            if g is not None and getattr(g, 'last_replayed', False):
                enc.slots.append((g.last_slot, g.last_slot['generation']))

    def _pad_mask(self, ids):
        """Key-padding mask of the article (:347).  On the GPU as the uint8 the attention kernels read - produced once,
        inside the encoder graph - instead of a bool that every decoder call converts."""
        m = ids == self.padding_idx
        return m.to(torch.uint8) if ids.is_cuda else m

    def _no_mask(self, B, P, device):
        """The all-visible mask of the image regions (:371): a constant, built once per shape."""
        cache = self.__dict__.setdefault('_no_mask_cache', {})
        key = (B, P, str(device))
        if key not in cache:
            cache[key] = torch.zeros(B, P, dtype=torch.uint8 if torch.device(device).type == 'cuda' else torch.bool,
                                     device=device)
        return cache[key]

    # ---- :311-397 -----------------------------------------------------------------
    def _forward(self, context, image, caption, face_embeds=None, obj_embeds=None, encoded=None):
        dtype = ops.rt.compute_dtype()
        cap = caption[self.index]
        target_ids = cap[:, 1:].contiguous()                               # :321-328
        caption_ids = cap[:, :-1].contiguous()
        caption[self.index] = caption_ids                                  # :329

        enc = encoded if encoded is not None else self.encode(context, image)   # frozen encoders (config :150-152)
        enc.wait()
        stack, x_image, article_mask = enc.stack, enc.x_image, enc.article_mask
        B, P, _ = x_image.shape
        ops.rt.wait_weight_update()      # everything below reads trainable weights (the encoders above do not)
        if self.weigh_bert:
            x_article = ops.mix_layers(stack, self.bert_weight)            # :355-364
        else:
            x_article = stack[-1]
        contexts = {
            'image': x_image.transpose(0, 1),
            'image_mask': self._no_mask(B, P, image.device),                  # :371
            'article': x_article.transpose(0, 1),
            'article_mask': article_mask,
        }
        if self.EXTRA_CONTEXTS:                                            # :373-379
            for key, emb in (('faces', face_embeds), ('obj', obj_embeds)):
                if key not in self.EXTRA_CONTEXTS:
                    continue
                Bf, n, dim = emb.shape
                if n == 0 or dim == 0:
                    contexts[key] = emb.new_zeros(n, Bf, dim).to(dtype)
                    contexts[key + '_mask'] = torch.zeros(Bf, n, dtype=torch.bool, device=emb.device)
                    continue
                clean = torch.empty(Bf, n, dim, dtype=dtype, device=emb.device)
                mask = torch.empty(Bf, n, dtype=torch.uint8, device=emb.device)
                ops.call('tell_nan_rows', emb.float().contiguous(), Bf * n, dim, clean, ops.hip.dt(dtype), mask)
                contexts[key] = clean.transpose(0, 1)
                contexts[key + '_mask'] = mask if mask.is_cuda else mask.bool()    # (uint8 is what the kernels read)
        return caption_ids, target_ids, contexts

    # ---- :67-140 ------------------------------------------------------------------
    def forward(self, context, image, caption, face_embeds=None, obj_embeds=None, metadata=None, names=None,
                attn_idx=None, encoded=None):
        """encoded: optional EncodedBatch of THIS batch produced earlier by `encode(..., ahead=True)`."""
        output_dict, caption_ids, contexts = self._forward_loss(context, image, caption, face_embeds, obj_embeds, encoded)
        if not self.training and self.evaluate_mode:                       # :92-116
            _, gen_ids, attns = self._generate(caption_ids, contexts, beam_size=getattr(self, 'eval_beam_size', 1))
            self._forward_generated(output_dict, gen_ids, attns, metadata)
        self.n_samples += caption_ids.shape[0]
        self.n_batches += 1
        return output_dict

    def _forward_loss(self, context, image, caption, face_embeds=None, obj_embeds=None, encoded=None):
        """:67-88 - encoders, teacher-forced decoder pass, loss in bits per token -> (output_dict, caption_ids, contexts)."""
        caption_ids, target_ids, contexts = self._forward(context, image, caption, face_embeds, obj_embeds, encoded)
        decoder_out = self.decoder(caption, contexts)
        loss_sum, sample_size = self.criterion(self.decoder.adaptive_softmax, decoder_out, target_ids)
        loss = ops.loss_bits(loss_sum, sample_size)                        # :85-88, bits per token
        return {'loss': loss, 'sample_size': sample_size.reshape(())}, caption_ids, contexts

    def _forward_generated(self, output_dict, gen_ids, attns, metadata):
        """:92-116 - what evaluate mode adds to the output once the captions are decoded: ids, text, per-sample BLEU."""
        ids_cpu = gen_ids.cpu()
        output_dict['gen_ids'] = ids_cpu.numpy()
        output_dict['attns'] = attns
        gen_texts = [self.detokenize(x[x > 1]) for x in ids_cpu]        # :96 "we ignore <s> and <pad>"
        output_dict['generations'] = gen_texts
        if metadata is not None:
            captions = [m.get('caption') or '' for m in metadata]
            output_dict['captions'] = captions
            output_dict['metadata'] = metadata
            import re
            from ..metrics import BleuScorer
            gens = [re.sub(r'[^\w\s]', '', t) for t in gen_texts]       # :105-106 remove punctuation
            refs = [re.sub(r'[^\w\s]', '', t) for t in captions]
            for gen, ref in zip(gens, refs):                            # :108-116
                scorer = BleuScorer(n=4)
                scorer += (gen, [ref])
                score, _ = scorer.compute_score(option='closest')
                for k in range(4):
                    self.sample_history['bleu-%d' % (k + 1)] += score[k] * 100

    def detokenize(self, ids):
        """`self.roberta.decode(ids)` of the reference (:96): BPE ids -> text.  Uses the encoder's own `decode` when it
        has one (a fairseq hub model), else the byte-level BPE of the data plane when its files are installed
        (data/indexers.bpe_directory), else the ids themselves as space-separated words (synthetic data has no text)."""
        dec = getattr(self.roberta, 'decode', None)
        if callable(dec):
            try:
                return dec(ids)
            except Exception:                                   # noqa: BLE001 - fall through to the local tokenizer
                pass
        bpe = self.__dict__.get('_bpe')
        if bpe is None:
            from ..data.bpe import RobertaBPE
            from ..data.indexers import bpe_directory
            try:
                bpe = RobertaBPE(bpe_directory())
            except FileNotFoundError:
                bpe = False
            self.__dict__['_bpe'] = bpe
        if bpe:
            return bpe.decode(ids)
        return ' '.join(str(int(i)) for i in ids if int(i) != 2)

    def generate(self, context, image, caption, face_embeds=None, obj_embeds=None, metadata=None, names=None,
                 attn_idx=None, beam_size=1, encoded=None):
        """encoded: optional EncodedBatch of THIS batch produced earlier by `encode(..., ahead=True)`."""
        caption_ids, _, contexts = self._forward(context, image, caption, face_embeds, obj_embeds, encoded)
        log_probs, gen_ids, attns = self._generate(caption_ids, contexts, attn_idx, beam_size=beam_size)
        return {'gen_ids': gen_ids, 'log_probs': log_probs, 'attns': attns}

    def lanes_usable(self):
        """Whether `generate_lanes` has a decode loop to interleave: the K/V-cached static-batch generator on a GPU."""
        return (self.fast_generation and hasattr(self.decoder, 'project_contexts')
                and next(self.parameters()).is_cuda)

    @torch.no_grad()
    def generate_lanes(self, batches, beam_size=1, lanes=2, forward=False):
        """Captions for a sequence of batches with `lanes` decode loops IN FLIGHT TOGETHER, each on its own stream with its
        own captured step, static buffers and counters (_decode_stepper(lane=)): a decode step is a chain of ~40 dependent
        launches that each fill the chip for a few microseconds and then wait on memory - at 12-27 % of the HBM roofline a second
        chain fits beside the first.  The host alternates the lanes' graph replays (one replay per lane and token).  The
        encoders of a group of batches run first (eval mode: no randomness, results identical to `generate`).
        Yields (batch, output) in order.  forward=True: the outputs of `forward` in evaluate mode (loss + captions + per-sample
        BLEU bookkeeping: what commands/evaluate.py consumes) instead of `generate`'s; beam_size then is `eval_beam_size`."""
        if forward:
            beam_size = getattr(self, 'eval_beam_size', 1)
            if self.training or not self.evaluate_mode or not self.lanes_usable():
                yield from self.generate_stream(batches, forward=True)  # nothing to decode / no static-batch decode loop
                return
        it = iter(batches)
        main = torch.cuda.current_stream()
        lane_streams = [streams.get('decode_lane_%d' % i) for i in range(lanes)]
        while True:
            group = []
            for _ in range(lanes):
                b = next(it, None)
                if b is not None:
                    group.append(b)
            if not group:
                return
            gens, outs, heads = [], [None] * len(group), []
            for ln, b in enumerate(group):
                f = {k: v for k, v in b.items() if k in ('context', 'image', 'caption', 'face_embeds', 'obj_embeds')}
                if forward:
                    od, caption_ids, contexts = self._forward_loss(**f)
                    heads.append((od, b.get('metadata'), caption_ids.shape[0]))
                else:
                    caption_ids, _, contexts = self._forward(**f)
                ev = torch.cuda.Event()
                ev.record(main)
                lane_streams[ln].wait_event(ev)
                with torch.cuda.stream(lane_streams[ln]), ops.hip.bound_stream():
                    g = (self._beam_steps(caption_ids, contexts, int(beam_size), lane=ln) if beam_size > 1 else
                         self._greedy_steps(caption_ids, contexts, lane=ln))
                gens.append(g)
            live = list(range(len(group)))
            while live:
                for ln in list(live):
                    with torch.cuda.stream(lane_streams[ln]), ops.hip.bound_stream():
                        try:
                            next(gens[ln])
                        except StopIteration as done:
                            lp, ids, attns = done.value
                            outs[ln] = {'gen_ids': ids, 'log_probs': lp, 'attns': attns}
                            live.remove(ln)
            for ln in range(len(group)):
                ev = torch.cuda.Event()
                ev.record(lane_streams[ln])
                main.wait_event(ev)
            for ln, (b, o) in enumerate(zip(group, outs)):
                if forward:
                    od, metadata, n = heads[ln]
                    self._forward_generated(od, o['gen_ids'], o['attns'], metadata)
                    self.n_samples += n
                    self.n_batches += 1
                    o = od
                yield b, o

    def generate_stream(self, batches, beam_size=1, forward=False):
        """Captions for a sequence of batches (the test-set loop of tell/commands/evaluate.py:118-160) with the frozen
        encoders of batch N+1 launched on their own streams BEFORE the decode loop of batch N is issued - the trainer's
        schedule applied to generation.  Numerics are those of `generate` / `forward` batch by batch: the encoders read
        nothing the decode loop writes, and consecutive batches' encoder outputs live in different buffer sets.

        MEASURED (MI355X, full model, 32 captions x 100 steps): 534 -> 544 captions/s greedy, 375 -> 381 beam 4 - not the
        25 % the two legs' sum promises.  A decode step is ~44 dependent launches of 5-20 us whose workgroups fill the
        chip for one round each; every one of them queues behind a wave of 100-us GEMM workgroups of the encoders, so
        the two mostly take turns.  Giving the decode chain compute units of its own (hipExtStreamCreateWithCUMask
        streams - hipGraph replays launched on them DO keep the mask, tools/probes/cumask_probe.hip) was built and
        measured: 64 / 128 CUs for the chain -> 260 / 416 captions/s - the chain's kernels are one round of
        latency-bound workgroups on 256 CUs and become two / four rounds on fewer (profiles/r05_cu_partition.txt); not
        kept.  What moves generation throughput is the batch (the iterator's knob): 544 / 747 / 939 captions/s greedy at
        32 / 64 / 128 captions per batch.

        batches: any iterable of batch dicts (keys of `forward`); forward=True yields `self(**batch)` (evaluate mode:
        loss + generation + metrics) instead of `generate(**batch)`.  Yields (batch, output_dict) in order."""
        it = iter(batches)
        cur = next(it, None)
        enc = None
        while cur is not None:
            nxt = next(it, None)
            ahead = None
            can = hasattr(self, 'encode') and torch.is_tensor(cur.get('image')) and cur['image'].is_cuda
            if can and enc is None:
                enc = self.encode(cur['context'], cur['image'])
            if can and nxt is not None and torch.is_tensor(nxt.get('image')) and nxt['image'].is_cuda:
                ahead = self.encode(nxt['context'], nxt['image'], ahead=True)
            if enc is not None and enc.stale():
                enc = self.encode(cur['context'], cur['image'])
            extra = {'encoded': enc} if enc is not None else {}
            out = self(**cur, **extra) if forward else self.generate(**cur, beam_size=beam_size, **extra)
            yield cur, out
            cur, enc = nxt, ahead

    # ---- :399-494 -----------------------------------------------------------------
    fast_generation = True      # projected-K/V cache + static batch; False = the reference's control flow

    def _generate(self, caption_ids, contexts, attn_idx=None, gen_len=100, eos=2, beam_size=1):
        if not hasattr(self.decoder, 'project_contexts'):
            # a recurrent decoder behind this model class (expt/*/3_lstm_roberta: `lstm_decoder_flattened`): greedy
            # decode that carries the LSTM state.  (The reference's loop feeds such a decoder only the last token with
            # an incremental_state its LSTMDecoder ignores, i.e. every step restarts from the initial state -
            # transformer_flattened.py:_generate with decoder_flattened_lstm.py:131-152; not reproduced.)
            from .baseline_glove import BaselineGloveModel
            lps, ids = BaselineGloveModel._generate(self, caption_ids, contexts, gen_len, eos)
            return lps, ids, []
        if beam_size > 1:
            return self._generate_beam(caption_ids, contexts, beam_size, gen_len, eos)
        if self.fast_generation:
            return self._generate_cached(caption_ids, contexts, gen_len, eos)
        return self._generate_reference_flow(caption_ids, contexts, attn_idx, gen_len, eos)

    @staticmethod
    def _drive(gen):
        """Run a decode generator (one `yield` per issued step) to its result."""
        try:
            while True:
                next(gen)
        except StopIteration as done:
            return done.value

    @torch.no_grad()
    def _generate_cached(self, caption_ids, contexts, gen_len=100, eos=2, check_every=8, lane=0):
        return self._drive(self._greedy_steps(caption_ids, contexts, gen_len, eos, check_every, lane))

    def _greedy_steps(self, caption_ids, contexts, gen_len=100, eos=2, check_every=8, lane=0):
        """A generator: yields after every issued decode step (generate_lanes interleaves two of these on two streams), returns
        (log_probs, ids, []).  Same greedy decode, restructured for the GPU: (1) context K/V projected once per caption,
        (2) the batch keeps its shape - finished rows are masked instead of compacted, so there is no
        per-step gather of the contexts and no per-step host synchronisation (the all-finished test
        runs every `check_every` steps), (3) fused arg-max over the adaptive softmax.  Rows are
        independent, so every row sees exactly the arithmetic of the reference flow: token ids are
        identical, pad=1 after EOS, output length = 1 + steps until the last row finished."""
        dec = self.decoder
        B = caption_ids.shape[0]
        dev = caption_ids.device
        kv = dec.project_contexts(contexts)
        step = self._decode_stepper(B, kv, contexts, gen_len, lane=lane)
        cur = caption_ids[:, 0:1].contiguous()
        finished = cur[:, 0] == eos
        fused = caption_ids.is_cuda and hasattr(step, 'cur')      # one bookkeeping launch per token (tell_greedy_update)
        if fused:
            # the histories live in STATIC buffers of the stepper (the bookkeeping launch is part of the captured step)
            bk = step.book('greedy', lambda: dict(
                ids=torch.empty(B, gen_len + 1, dtype=torch.long, device=dev),
                lps=torch.empty(B, gen_len, dtype=torch.float32, device=dev),
                done_step=torch.empty(B, dtype=torch.long, device=dev), fin8=torch.empty(B, dtype=torch.uint8, device=dev)))
            ids, lps, done_step, fin8 = bk['ids'], bk['lps'], bk['done_step'], bk['fin8']
            ids.fill_(self.padding_idx)
            lps.zero_()
            done_step.fill_(gen_len)
        else:
            ids = torch.full((B, gen_len + 1), self.padding_idx, dtype=torch.long, device=dev)
            lps = torch.zeros(B, gen_len, dtype=torch.float32, device=dev)
            done_step = torch.full((B,), gen_len, dtype=torch.long, device=dev)   # steps this row took part in
        ids[:, 0] = cur[:, 0]
        done_step[finished] = 0
        steps = gen_len
        if fused:
            fin8.copy_(finished)
            step.cur.copy_(cur)
            inv_temp = 1.0 / float(self.sampling_temp)

            def book(out, i, step_dev):
                # the step's LAST launch: inside the captured step it takes the step index from the device counter
                # (step_dev), eagerly from the host; either way it leaves the next step's position offset behind
                tok, lp = out
                ops.call('tell_greedy_update', tok.reshape(B), lp.reshape(B), fin8, ids, ids.stride(0), lps, lps.stride(0),
                         done_step, step.cur, B, int(i), int(eos), inv_temp, step.counter_out, step_dev)
        i = 0
        while fused and i < gen_len:
            # (the bookkeeping launch of step i - 1 left the position offset of step i in the device counter: no fill launch;
            #  the host's part of a step is ONE graph replay - and from the first all-finished test on, of `check_every` steps)
            n = check_every if (i >= check_every and i % check_every == 0 and i + check_every <= gen_len) else 1
            if n == 1 or not step.multi(i, n, book):
                n = 1
                step(i, None, counter_set=i > 0, post=book)
            i += n
            yield i - 1
            if i % check_every == 0 and bool(fin8.all()):
                break
        for i in range(0 if fused else gen_len):
            tok, lp = step(i, cur)
            yield i
            tok = tok.long().view(B)
            lp = lp.view(B) / self.sampling_temp
            ids[:, i + 1] = torch.where(finished, ids[:, i + 1], tok)
            lps[:, i] = torch.where(finished, lps[:, i], lp)
            newly = (~finished) & (tok == eos)
            done_step = torch.where(newly, torch.full_like(done_step, i + 1), done_step)
            finished = finished | newly
            cur = tok.view(B, 1)
            if (i + 1) % check_every == 0 and bool(finished.all()):
                break
        steps = int(done_step.max())                                          # one sync at the end
        steps = max(steps, 1)
        if fused:                                                             # (the static buffers belong to the stepper)
            return lps[:, :steps].clone(), ids[:, :steps + 1].clone(), []
        return lps[:, :steps], ids[:, :steps + 1], []

    def _decode_stepper(self, B, kv, contexts, gen_len, topk=0, lane=0):
        """-> step(i, cur [B,1]) -> (token [B,1], log-prob [B,1]) - or, with topk=k, the k best (tokens [B,1,k],
        log-probs [B,1,k]) of every row - for the cached greedy / beam generators; step.reorder(rows) permutes the
        rows of the incremental state (beam search).

        With graphs enabled the decode step (about 150 launches of a few microseconds each, host-bound when issued
        one by one) is captured ONCE per (batch, context shapes) signature and replayed: every tensor it touches is
        static - the DynamicConv input buffers have their final K-1 rows from the start (zero history), the
        projected K/V and masks are copied into fixed buffers per caption batch, and the position offset comes from
        the graph's device step counter (embed_finalize reads it, like the dropout kernels)."""
        dec = self.decoder
        names = [n for layer_kv in kv[:1] for n in layer_kv]
        if not graphs.ENABLED or self.training or not torch.is_tensor(kv[0][names[0]][0]) or \
                not kv[0][names[0]][0].is_cuda:
            state = {}
            head = (lambda x: dec.adaptive_softmax.topk(x, topk)) if topk else dec.adaptive_softmax.greedy

            def eager_step(i, cur):
                return head(dec({self.index: cur}, contexts, incremental_state=state, kv_cache=kv)[0][:, -1:])
            eager_step.reorder = lambda rows: dec.reorder_incremental_state(state, rows)
            return eager_step
        dev, dtype = kv[0][names[0]][0].device, kv[0][names[0]][0].dtype
        # lane: decode loops that are in flight TOGETHER (generate_lanes: two caption batches decoded on two streams) own
        # their graphs, static buffers, counters and split-reduction workspace
        sig = (B, dtype, topk, int(gen_len), tuple((n, tuple(kv[0][n][0].shape), tuple(kv[0][n][1].shape)) for n in names),
               dec.embedder.token_embedder_position.weights.data_ptr(), int(lane))
        cache = self.__dict__.setdefault('_decode_graphs', {})
        # A captured step bakes in the addresses of the working weights (weight-normalised copies, the concatenated
        # softmax head) that ops._cached rebuilds - at NEW addresses - whenever the weights change (optimizer step,
        # load_state_dict): every capture belongs to one state of the weights and is dropped with it
        stamp = (ops.rt.weights_epoch(), sum(p._version for p in dec.parameters()))
        if self.__dict__.get('_decode_graphs_stamp') != stamp:
            cache.clear()
            self.__dict__['_decode_graphs_stamp'] = stamp
        h = cache.get(sig)
        if h is None:
            if len(cache) >= graphs.MAX_SIGNATURES:
                cache.pop(next(iter(cache)))
            h = cache[sig] = {
                # counter[0]: the position offset the kernels of a replay read; counter[1]: the NEXT step's offset when
                # the bookkeeping launch is part of the captured step (`ig`, below)
                'graph': None, 'counter': torch.zeros(2, dtype=torch.int32, device=dev), 'book': {},
                'cur': torch.zeros(B, 1, dtype=torch.long, device=dev),
                'kv': None,
                # (key-padding masks as the uint8 the attention kernels read: converted once per caption batch)
                'ctx': {k: torch.empty_like(v, dtype=torch.uint8 if v.dtype == torch.bool else v.dtype)
                        for k, v in contexts.items() if torch.is_tensor(v)},
                'state': dec.static_incremental_state(B, dev, dtype, beam=bool(topk)),
            }
            po = dec.embedder.token_embedder_position            # the table must already cover the longest caption
            po.next_start(gen_len + 2, None)
            # The static copy of the projected K / V.  Where the weight-streaming step takes this batch (decode.usable: all
            # four attentions of a layer are one tell_attn_decode launch) the copy is HEAD-MAJOR - [B, H, S, 64], handed on
            # as [S, B, H, 64] views: the keys a (sample, head) workgroup walks are one contiguous block instead of 128-byte
            # pieces a whole [B, 2E] projection row (128 KB at B = 32) apart.  The re-layout rides on the copy into the
            # static buffers that the captured step needs anyway, once per caption batch.
            from .. import decode as _dec
            probe = torch.empty(1, B, dec.embedder.get_output_dim(), dtype=dtype, device=dev)
            hm = _dec.KV_HEAD_MAJOR and dtype == torch.bfloat16 and _dec.usable(dec, probe, h['state'], kv)

            def static_like(t, mod):
                if hm and t.shape[0] > 0 and t.dim() == 3 and t.shape[2] == mod.num_heads * 64:
                    S_, Bc, H_ = t.shape[0], t.shape[1], mod.num_heads
                    return torch.empty(Bc, H_, S_, 64, dtype=t.dtype, device=t.device).permute(2, 0, 1, 3)
                return torch.empty_like(t)
            # (measured, B = 32: packed 28.6 -> 22.6-24.9 us per launch at beam 4; with ONE hypothesis per sample the VALU kernel on
            #  the head-major cache is the faster one, 18.7 against 20.4 us - packed from two hypotheses per sample on)
            n_cached = kv[0][names[0]][0].shape[1] if kv[0][names[0]][0].dim() == 3 else B
            several = B >= _dec.PACKED_MIN_HYP * max(int(n_cached), 1)
            layer_pk = (B > _dec.MAX_ROWS and dtype == torch.bfloat16 and bool(h['state'].get('_ring')) and
                        _dec.layer_path_takes_packed(dec))     # (the layer-by-layer step above MAX_ROWS rows)
            if _dec.KV_PACKED and several and (hm or layer_pk):
                # ... or PACKED for the matrix cores: keys head-major with the two virtual keys appended, values transposed
                # and permuted (decode.PackedKV); one launch per layer reads all four contexts (tell_attn_decode_packed)
                h['kv'] = [{n: _dec.PackedKV(layer.context_attns[n], pair[0].shape[0], pair[0].shape[1], dev)
                            for n, pair in lk.items()} for lk, layer in zip(kv, dec.layers)]
            else:
                h['kv'] = [{n: tuple(static_like(t, layer.context_attns[n]) for t in pair) for n, pair in lk.items()}
                           for lk, layer in zip(kv, dec.layers)]
            # in-graph bookkeeping needs the step's first kernel to be tell_embed_gather_step (it publishes the counter)
            h['ig'] = bool(_dec.IN_GRAPH_BOOK and dtype == torch.bfloat16 and _dec.usable(dec, probe, h['state'], kv) and
                           _dec.embed_usable(dec.embedder, h['cur'], h['state']))
        for lk, ls in zip(kv, h['kv']):
            for n, pair in lk.items():
                if not isinstance(ls[n], tuple):                  # decode.PackedKV
                    mk = contexts.get(n + '_mask')
                    ls[n].fill(pair[0], pair[1], mk)
                    continue
                for t, s in zip(pair, ls[n]):
                    s.copy_(t.view(s.shape) if s.dim() == 4 else t)
        for k, s in h['ctx'].items():
            s.copy_(contexts[k])
        dec.reset_static_state(h['state'])
        pos_key = dec.embedder.token_embedder_position._state_key
        h['state'].pop(pos_key, None)

        head = (lambda x: dec.adaptive_softmax.topk(x, topk)) if topk else dec.adaptive_softmax.greedy

        c_cur, c_next = h['counter'][0:1], h['counter'][1:2]
        # where a bookkeeping launch leaves the next step's offset: the word the embedder's kernel reads (in-graph
        # bookkeeping), or the counter itself
        c_out = c_next if h['ig'] else c_cur

        def run():
            from .. import decode as _dec2
            prev_lane = _dec2.CUR_LANE[0]
            _dec2.CUR_LANE[0] = int(lane)
            try:
                out = dec({self.index: h['cur']}, h['ctx'], incremental_state=h['state'], kv_cache=h['kv'])
                return head(out[0][:, -1:])
            finally:
                _dec2.CUR_LANE[0] = prev_lane

        def eager(i, post):
            res = run()
            if post is not None:
                post(res, i, None)
            return res

        def step(i, cur, counter_set=False, post=None):
            """post(out, i, step_dev): the caller's per-token bookkeeping launch (over static buffers: step.book).  With
            in-graph bookkeeping it is recorded as the LAST launch of the captured step."""
            if cur is not None:                                   # (None: the caller already wrote step.cur)
                h['cur'].copy_(cur)
            if h['graph'] is None and i != 1:
                return eager(i, post)                             # warm step(s) before the capture, or fallback
            if h['graph'] is None:                                # i == 1: the host position state is 1 now
                inside = post is not None and h['ig']
                # the host's part of the step's position state as THIS capture sees it (step.multi records further steps
                # with the same constants: the device counter is what moves a recorded step along)
                h['host_ints'] = {k_: v_ for k_, v_ in h['state'].items() if isinstance(v_, int) and not isinstance(v_, bool)}
                try:
                    g = torch.cuda.CUDAGraph()
                    try:
                        ops.call('tell_set_rng_step_ptr', c_cur)
                        ops.call('tell_set_pos_step_ptr', c_cur)
                        if inside:
                            ops.call('tell_set_pos_next_ptr', c_next)
                        # (the captured step's resident GEMM launches keep their tile-counter slots until this entry is
                        #  dropped - `held` gives them back, like StepGraph / GraphedCall do)
                        with graphs.no_gc(), ops.hip.tile_slots() as held, torch.cuda.graph(g):
                            with ops.hip.bound_stream():
                                h['out'] = run()
                                if inside:
                                    post(h['out'], i, c_cur)
                        h['tile_slots'] = held
                    finally:
                        ops.call('tell_set_rng_step_ptr', None)
                        ops.call('tell_set_pos_step_ptr', None)
                        ops.call('tell_set_pos_next_ptr', None)
                    h['graph'], h['base'], h['graph_has_post'] = g, 1, inside
                except Exception as exc:                          # noqa: BLE001 - stay eager for this signature
                    h['graph'], h['error'] = False, repr(exc)
                    return eager(i, post)
            if h['graph'] is False:
                return eager(i, post)
            if not counter_set:
                # position offset of this step (may be -1): into the word the step's first kernel reads
                (c_next if h.get('graph_has_post') else c_cur).fill_(i - h['base'])
            h['graph'].replay()
            if post is not None and not h.get('graph_has_post'):
                post(h['out'], i, None)
            return h['out']

        def multi(i, n, post):
            """Steps i .. i + n - 1 as ONE graph replay (n consecutive steps recorded into one graph: the bookkeeping launch
            that ends a recorded step leaves the position offset of the next one in the device counter, so the steps chain
            on the device exactly as n single replays would - what goes is the per-replay cost between them, ~25 us of a
            360-490 us step).  Needs the single-step graph with in-graph bookkeeping (captured at step 1) and the counter
            already set by step i - 1's bookkeeping launch.  -> False: not available, issue the steps one by one."""
            if not h.get('graph') or not h.get('graph_has_post') or post is None or i < 2 or not MULTI_STEP_GRAPHS:
                return False
            key = ('multi', int(n))
            g = h.get(key)
            if g is None:
                saved = {k_: h['state'].get(k_) for k_ in h['host_ints']}
                try:
                    g = torch.cuda.CUDAGraph()
                    try:
                        ops.call('tell_set_rng_step_ptr', c_cur)
                        ops.call('tell_set_pos_step_ptr', c_cur)
                        ops.call('tell_set_pos_next_ptr', c_next)
                        with graphs.no_gc(), ops.hip.tile_slots() as held, torch.cuda.graph(g):
                            with ops.hip.bound_stream():
                                for j in range(int(n)):
                                    h['state'].update(h['host_ints'])   # the constants of the single-step capture
                                    post(run(), i + j, c_cur)
                        h[key + ('slots',)] = held
                    finally:
                        ops.call('tell_set_rng_step_ptr', None)
                        ops.call('tell_set_pos_step_ptr', None)
                        ops.call('tell_set_pos_next_ptr', None)
                        h['state'].update(saved)
                    h[key] = g
                except Exception as exc:                          # noqa: BLE001 - keep the single-step replays
                    h[key], h['multi_error'] = False, repr(exc)
                    return False
            if g is False:
                return False
            g.replay()
            return True

        def book(kind, make):
            if kind not in h['book']:
                h['book'][kind] = make()
            return h['book'][kind]

        def reorder(rows, group=0):                               # in place: the buffers are part of the graph
            if h['state'].get('_ring'):                           # rings: only the ancestor table changes
                dec.reorder_incremental_state(h['state'], rows)
                return
            bufs = [s for k_, s in h['state'].items() if 'Conv1dTBC' in k_ and torch.is_tensor(s) and s.shape[0] > 0]
            if (group and 1 <= group <= 8 and bufs and all(s.dtype == torch.bfloat16 and s.is_contiguous() and
                                                           s.shape[2] == 1024 for s in bufs) and len(bufs) <= 8):
                # rows[r] lies inside r's group of `group` hypotheses: every layer's buffer in ONE launch
                ops.call('tell_reorder_rows', len(bufs), ops._ptr_array(bufs), ops._int_array([s.shape[0] for s in bufs]),
                         rows, bufs[0].shape[1], 1024, int(group))
                return
            for s in bufs:
                s.copy_(s.index_select(1, rows))
        step.reorder = reorder
        step.multi = multi
        step.cur = h['cur']
        step.book = book
        step.counter_out = c_out                                  # (base 1: the offset of step i is i - 1)
        step.back = h['state'].get('_back')                       # ancestor table of the DynamicConv rings, or None
        return step

    @torch.no_grad()
    def _generate_beam(self, caption_ids, contexts, beam_size, gen_len=100, eos=2, check_every=8, lane=0):
        return self._drive(self._beam_steps(caption_ids, contexts, beam_size, gen_len, eos, check_every, lane))

    def _beam_steps(self, caption_ids, contexts, beam_size, gen_len=100, eos=2, check_every=8, lane=0):
        """A generator like _greedy_steps.  Beam search on the cached static-shape generator (SURVEY 8-f1 / BASELINE config 5; the reference itself
        only samples top-1, transformer_faces_objects.py:443-464).  B*K rows (row = b*K + j) stay resident; the
        projected K/V of the static contexts are computed once per caption and replicated per beam; the DynamicConv
        input buffers are reordered by parent with the reference's `reorder_incremental_state` contract
        (dynamic.py:338-342).  Score = sum of token log-probs (no length penalty); a finished hypothesis keeps its
        score and is extended with pad only.  -> (log_probs [B,steps], ids [B,steps+1] of the best hypothesis, [])."""
        dec = self.decoder
        B, K = caption_ids.shape[0], int(beam_size)
        dev = caption_ids.device
        pad = self.padding_idx
        rep = lambda t, dim: t.repeat_interleave(K, dim=dim).contiguous()           # noqa: E731
        # contexts, masks and projected K/V stay at batch B: the attention modules present the K hypotheses of a
        # sample as K query positions of that sample (modules/attention.py), nothing is replicated per beam
        ctx = {k_: v_ for k_, v_ in contexts.items() if torch.is_tensor(v_)}
        kv = dec.project_contexts(contexts)
        step = self._decode_stepper(B * K, kv, ctx, gen_len, topk=K, lane=lane)
        cur = rep(caption_ids[:, 0:1], 0)
        finished = (cur[:, 0] == eos).view(B, K)
        fused = caption_ids.is_cuda and hasattr(step, 'cur') and K <= 8 and gen_len + 1 <= 256
        if fused:
            bk = step.book('beam', lambda: dict(
                cum=torch.empty(B, K, dtype=torch.float32, device=dev), fin8=torch.empty(B, K, dtype=torch.uint8, device=dev),
                seqs=torch.empty(B, K, gen_len + 1, dtype=torch.long, device=dev),
                lps=torch.empty(B, K, gen_len, dtype=torch.float32, device=dev),
                rows=torch.empty(B * K, dtype=torch.long, device=dev)))
            cum, fin8, seqs, lps, rows = bk['cum'], bk['fin8'], bk['seqs'], bk['lps'], bk['rows']
            cum.fill_(float('-inf'))
            seqs.fill_(pad)
            lps.zero_()
        else:
            cum = torch.full((B, K), float('-inf'), dtype=torch.float32, device=dev)
            seqs = torch.full((B, K, gen_len + 1), pad, dtype=torch.long, device=dev)
            lps = torch.zeros(B, K, gen_len, dtype=torch.float32, device=dev)
        cum[:, 0] = 0.0                                     # all K rows start identical: only hypothesis 0 counts
        seqs[:, :, 0] = cur.view(B, K)
        base = (torch.arange(B, device=dev) * K).view(B, 1)
        n_steps = gen_len
        if fused:
            # one bookkeeping launch per token (tell_beam_update: candidate scores, top-K per sample, histories gathered
            # by parent, next inputs; ring buffers: it also composes the ancestor table with this step's parents - no row
            # of any layer's DynamicConv buffer is moved) - the LAST launch of the captured step where the stepper allows
            fin8.copy_(finished)
            step.cur.copy_(cur)
            ring = step.back is not None
            inv_temp = 1.0 / float(self.sampling_temp)

            def book(out, i, step_dev):
                tk, lp = out
                ops.call('tell_beam_update', tk, lp, cum, fin8, seqs, lps, step.cur, rows, B, K, gen_len + 1, int(i), int(pad),
                         int(eos), inv_temp, step.back, step.back.shape[0] if ring else 0, step.counter_out, step_dev)
            i = 0
            while i < gen_len:
                # (time-ordered buffers - fp32 parity mode: the bookkeeping stays a host-side launch and one more launch
                #  re-orders every layer's rows by parent)
                n = check_every if (ring and i >= check_every and i % check_every == 0 and i + check_every <= gen_len) else 1
                if n == 1 or not step.multi(i, n, book):
                    n = 1
                    out = step(i, None, counter_set=i > 0, post=book if ring else None)
                    if not ring:
                        book(out, i, None)
                        step.reorder(rows, K)
                i += n
                yield i - 1
                if i % check_every == 0 and bool(fin8.all()):
                    n_steps = i
                    break
        for i in range(0 if fused else gen_len):
            # each hypothesis contributes its own K best tokens (the best K of K x V always lie among them)
            tk, lp = step(i, cur)
            tk, lp = tk.view(B, K, K).long(), lp.view(B, K, K) / self.sampling_temp
            # a finished hypothesis has ONE continuation: pad, at no cost
            fin = finished.unsqueeze(-1)
            first = torch.zeros(K, dtype=torch.bool, device=dev)
            first[0] = True
            lp = torch.where(fin, torch.where(first, torch.zeros_like(lp), torch.full_like(lp, float('-inf'))), lp)
            tk = torch.where(fin, torch.full_like(tk, pad), tk)
            top, idx = (cum.unsqueeze(-1) + lp).view(B, K * K).topk(K, dim=1)       # sorted, best first
            parent = idx // K
            tok = tk.view(B, K * K).gather(1, idx)
            rows = (base + parent).view(-1)
            was_finished = finished.gather(1, parent)
            tok = torch.where(was_finished, torch.full_like(tok, pad), tok)
            seqs = seqs.view(B * K, -1).index_select(0, rows).view(B, K, -1)
            lps = lps.view(B * K, -1).index_select(0, rows).view(B, K, -1)
            seqs[:, :, i + 1] = tok
            lps[:, :, i] = torch.where(was_finished, torch.zeros_like(top), top - cum.gather(1, parent))
            finished = was_finished | (tok == eos)
            cum = top
            step.reorder(rows)
            cur = tok.view(B * K, 1)
            yield i
            if (i + 1) % check_every == 0 and bool(finished.all()):
                n_steps = i + 1
                break
        best = seqs[:, 0]                                    # topk keeps hypotheses sorted by score
        steps = int((best[:, 1:] != pad).sum(1).max())       # one sync: length of the longest best caption
        steps = max(min(steps, n_steps), 1)
        if fused:                                            # (the static buffers belong to the stepper)
            return lps[:, 0, :steps].clone(), best[:, :steps + 1].clone(), []
        return lps[:, 0, :steps], best[:, :steps + 1], []

    @torch.no_grad()
    def _generate_reference_flow(self, caption_ids, contexts, attn_idx=None, gen_len=100, eos=2):
        """Greedy decoding with the reference's semantics (finished rows leave the batch, pad=1 after
        EOS, loop ends when no row is active).  The arg-max over the 50 265-way adaptive softmax is
        fused (no [B, vocab] log-prob tensor)."""
        state = {}
        B = caption_ids.shape[0]
        dev = caption_ids.device
        seed = caption_ids[:, 0:1]
        alive = seed[:, -1] != eos
        keep = alive
        cur = seed
        log_probs, paths, attns = [], [seed], []
        names = [k for k in contexts if not k.endswith('_mask') and not k.startswith('_')]
        for _ in range(gen_len):
            self.decoder.filter_incremental_state(state, keep)                      # :417
            ctx_i = {}
            for n in names:                                                         # :420-431
                ctx_i[n] = contexts[n][:, alive]
                ctx_i[n + '_mask'] = contexts[n + '_mask'][alive]
            dec_out = self.decoder({self.index: cur[:, -1:]}, ctx_i, incremental_state=state)
            attns.append(dec_out[1]['attn'])
            tok, lp = self.decoder.adaptive_softmax.greedy(dec_out[0][:, -1:])      # :443-464
            sel_ix = tok.long()
            sel_lp = lp / self.sampling_temp
            full_lp = sel_lp.new_zeros(B, 1)
            full_lp[alive] = sel_lp
            full_ix = sel_ix.new_full((B, 1), self.padding_idx)
            full_ix[alive] = sel_ix
            log_probs.append(full_lp)
            paths.append(full_ix)
            keep = sel_ix.squeeze(-1) != eos                                        # :476-483
            alive = alive.clone()
            alive[alive.nonzero().squeeze(1)[~keep]] = False
            cur = torch.cat([cur, sel_ix], dim=1)[keep]
            if int(keep.sum()) == 0:                                                # :485
                break
        return torch.cat(log_probs, dim=-1), torch.cat(paths, dim=-1), attns

    def get_metrics(self, reset=False):                                             # :504-517
        metrics = {'_n_batches': self.n_batches, '_n_samples': self.n_samples}
        for key, value in self.sample_history.items():
            metrics[key] = value / max(self.n_samples, 1)
        if reset:
            self.n_batches = 0
            self.n_samples = 0
            self.sample_history = defaultdict(float)
        return metrics


@Model.register('transformer_faces_objects')
class TransformerFacesObjectModel(CaptionModel):
    """tell/models/transformer_faces_objects.py:22-23"""
    USE_FACES_OBJECTS = True
    EXTRA_CONTEXTS = ('faces', 'obj')


@Model.register('transformer_faces')
class TransformerFacesModel(CaptionModel):
    """tell/models/transformer_faces.py:21-22 (expt/*/8_transformer_faces): image + article + faces, driven by
    `dynamic_conv_decoder_faces_parallel`; the faces+objects forward without `obj_embeds`."""
    USE_FACES_OBJECTS = True
    EXTRA_CONTEXTS = ('faces',)


@Model.register('transformer_flattened')
class TransformerFlattenedModel(CaptionModel):
    """tell/models/transformer_flattened.py:23-24 (also drives `dynamic_conv_decoder_flattened_no_image`)"""
    USE_FACES_OBJECTS = False
    EXTRA_CONTEXTS = ()
