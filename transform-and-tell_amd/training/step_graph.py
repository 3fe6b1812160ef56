"""The decoder half of the optimisation step - decoder forward, adaptive-softmax loss, backward and (single process)
BertAdam, callback_apex_trainer.py:208-247 minus the frozen encoders - as ONE hipGraph per shape signature.

Issued kernel by kernel this half is ~900 dependent launches of 5-30 us each (M = T*B = 1024 rows): 22 ms of Python
per step at BASELINE configs[2] against 27 ms of GPU time, so the host is the bound.  Captured, a step costs the host
a handful of small copies and one graph launch.

What makes the capture legal:
  * no host synchronisation anywhere in the step (device-side band compaction, device row counts, the NaN/Inf skip is
    a device flag read by the optimizer kernel, ops / csrc/optim.hip);
  * dropout draws fresh masks from the graph's device step counter (tell_set_rng_step_ptr, csrc/common.h);
  * weight gradients accumulate into the flat fp32 gradient buffer, parameters / moments / the bf16 shadow live in
    flat static buffers (training/optimizers.py) - every pointer a kernel sees is stable;
  * the weight-normalised working weights are rebuilt INSIDE the graph (ops._cached runs its makers while capturing);
  * the learning rate is computed on the device from the count of applied updates (csrc/optim.hip lr_schedule_kernel).

Inputs: the outputs of the frozen encoders are consumed in place when they are graph-owned static buffers (the
encoder GraphedCalls; one capture per buffer slot), copied into static buffers otherwise (stand-in encoders of the
tests); token ids and the face / object arrays are always copied (small).

Data parallel: the graph ends after backward; the token-count and gradient exchanges and the optimizer stay outside
(trainer._update).  The per-rank loss weight n_r * world / sum(n) is applied to the gradient on its way to the wire
(tell_scale_cast) instead of to the loss before backward - the same numbers, gradients are linear in that factor."""
import os

import torch

from .. import graphs, hip, ops
from .. import runtime as rt

ENABLED = os.environ.get('TELL_STEP_GRAPH', '1') != '0'
COPY_LIMIT = 256 << 20          # encoder outputs that are not graph-owned are copied in when smaller than this


class StepGraph:
    """Captured decoder steps, one per shape signature.  Real batches vary in shape (the iterator pads to the per-batch
    maximum): the trainer pads them to a small set of buckets first (Trainer.shape_buckets), a signature is only
    captured at its `capture_after`-th sighting, at most 4 * graphs.MAX_SIGNATURES = 32 captures are kept (least recently
    used evicted, which releases their activation pools) and the sighting counters themselves are bounded
    (graphs.SignatureCache)."""

    def __init__(self, trainer, capture_after=None):
        self.tr = trainer
        # ~3 GB of private activation pool per captured signature at B = 32 (bench leg: 6 graphs, 23.6 GB reserved in all):
        # 32 graphs are ~100 of the 288 GB.  Beyond that the cache's thrash guard (graphs.SignatureCache) decides.
        self.cache = graphs.SignatureCache(int(os.environ.get('TELL_STEP_GRAPHS_MAX', 4 * graphs.MAX_SIGNATURES)), capture_after)
        self.entries = self.cache.entries
        self.replays = 0
        # ONE memory pool for every captured signature of this trainer (round 5).  A private pool per graph pinned 3.9 GB per
        # signature (66.5 GB for the 17 signatures of bench.py's many-signature leg) although the graphs never run at
        # the same time: replays are serial on the training stream, a graph's intermediates are dead when its replay
        # ends, and what outlives it (loss, token count) is read before the next replay starts.  With a shared pool a new
        # capture reuses the blocks the earlier captures' intermediates returned.  TELL_STEP_POOL=0: private pools (A/B).
        self.shared_pool = os.environ.get('TELL_STEP_POOL', '1') != '0'
        self.pool = None

    def reset(self):
        self.cache.clear()

    # ------------------------------------------------------------------ signature
    def _plan(self, batch, enc):
        tr = self.tr
        model = tr.model
        idx = model.index
        if not (ENABLED and graphs.ENABLED) or not hasattr(model, 'encode') or not model.training:
            return None
        if tr.async_update:                                # the opt-in update stream belongs to the eager schedule
            return None
        cap = batch['caption'][idx]
        ctx = batch['context'][idx]
        if not (torch.is_tensor(cap) and cap.is_cuda):
            return None
        by_ptr = bool(getattr(enc, 'static', False))
        big = [enc.stack, enc.x_image]
        if not all(torch.is_tensor(t) for t in big):
            return None                                   # a user encoder returned a list: stay eager
        if not by_ptr and sum(t.numel() * t.element_size() for t in big) > COPY_LIMIT:
            return None
        small = {'ctx': ctx, 'cap': cap}
        for k in ('face_embeds', 'obj_embeds'):
            if batch.get(k) is not None:
                small[k] = batch[k]
        sig = (tuple((k, tuple(v.shape), v.dtype) for k, v in small.items()),
               tuple((tuple(t.shape), t.dtype, t.data_ptr() if by_ptr else 0) for t in big),
               by_ptr, rt.compute_dtype(), tr.dp, ops._WGRAD['enabled'], tr.defer_update,
               # the gradient convention - stores or zero + accumulate - is baked into the launches; the trainer's ONE
               # observing step runs zero + accumulate but is followed by captures in store mode: it counts as a sighting
               # of the store-mode signature
               ops.grad_store_on() or (tr._store == 'want' and not tr.defer_update))
        return sig, small, big, by_ptr

    # ------------------------------------------------------------------ one step
    def run(self, batch, enc, eager_step):
        """-> detached loss, or None when this batch cannot go through a graph (caller runs the eager step).
        eager_step(batch, enc) is the uncaptured step; a signature runs eagerly until its capture_after-th sighting
        (that call also builds every cache the capture relies on), the graph is recorded right after it and replayed
        from the next call on."""
        plan = self._plan(batch, enc)
        if plan is None:
            return None
        sig, small, big, by_ptr = plan
        e = self.cache.touch(sig)
        if e['state'] == 'seen':
            if self.cache.due(e) and e['hits'] >= 2:
                # this signature has run eagerly before (every cache a capture relies on exists): record the graph NOW and
                # train this batch with its first replay - a new signature costs one eager step + one capture pass, not two
                # eager steps + a capture (variable-length data: 12 signatures in the first epochs, 55 ms per eager step)
                if self.cache.make_room():              # (False: full and frozen by the thrash guard - stay eager)
                    self._capture(e, batch, small, big, by_ptr, enc)
                if e['state'] != 'ready':
                    return eager_step(batch, enc)
            else:
                loss = eager_step(batch, enc)
                if self.cache.due(e) and self.cache.make_room():
                    self._capture(e, batch, small, big, by_ptr, enc)
                return loss
        if e['state'] != 'ready':
            return None
        tr = self.tr
        enc.wait()                                         # the encoder streams join the current stream here
        for k, v in small.items():
            e['small'][k].copy_(v)
        if not by_ptr:
            for s, t in zip(e['big'], big):
                s.copy_(t)
        e['counter'].fill_(e['replays'])
        e['replays'] += 1
        self.replays += 1
        if not tr.dp:
            tr.optimizer.prepare()
        e['graph'].replay()
        model = tr.model
        model.n_samples += small['cap'].shape[0]
        model.n_batches += 1
        if tr.dp:
            tr._update(n_local=e['sample_size'])
        elif not tr.defer_update:
            tr.optimizer.advance()
            tr.flat.accum_pending = False               # (the captured update consumed and cleared the gradients)
        return e['loss'].clone()

    # ------------------------------------------------------------------ capture
    def _capture(self, e, batch, small, big, by_ptr, enc):
        from ..models.transformer import EncodedBatch
        tr = self.tr
        model = tr.model
        idx = model.index
        dev = small['cap'].device
        st_small = {k: v.clone() for k, v in small.items()}
        st_big = big if by_ptr else [t.clone() for t in big]
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        saved = (model.n_samples, model.n_batches, tr.optimizer.step_count)
        g = torch.cuda.CUDAGraph()
        held = hip.tile_slots()                 # (csrc/gemm.hip tile counters: released when this entry is dropped)
        tr._capturing = True
        try:
            ops.drop_trainable_cache()          # working weights of trainable parameters are rebuilt inside the graph
            hip.call('tell_set_rng_step_ptr', counter)
            if self.shared_pool and self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            with graphs.no_gc(), held, torch.cuda.graph(g, pool=self.pool if self.shared_pool else None,
                                                        capture_error_mode='thread_local'):
                with hip.bound_stream():
                    encs = EncodedBatch()
                    encs.stack, encs.x_image = st_big
                    encs.article_mask = st_small['ctx'] == model.padding_idx
                    kw = {k: st_small[k] for k in ('face_embeds', 'obj_embeds') if k in st_small}
                    out = model(context={idx: st_small['ctx']}, image=batch['image'], caption={idx: st_small['cap']},
                                encoded=encs, **kw)
                    loss = out['loss']
                    tr._flag_loss(loss)
                    tr._backward(loss)
                    if not tr.dp and not tr.defer_update:
                        tr.optimizer.launch(grad_scale=1.0, zero_grad=True, skip=tr.skip)
            e.update(state='ready', graph=g, small=st_small, big=st_big, counter=counter, replays=1,
                     loss=loss.detach(), sample_size=out['sample_size'], tile_slots=held)
        except Exception as exc:                # noqa: BLE001 - any capture problem -> this signature stays eager
            e['state'] = 'failed'
            e['error'] = repr(exc)
            if os.environ.get('TELL_STEP_GRAPH_STRICT') == '1':
                raise
        finally:
            tr._capturing = False
            hip.call('tell_set_rng_step_ptr', None)
            model.n_samples, model.n_batches, tr.optimizer.step_count = saved
            ops.drop_trainable_cache()          # entries made while capturing point into the graph's pool
            rt.bump_weights_epoch()
