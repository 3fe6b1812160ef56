"""The optimisation step of tell/training/callback_apex_trainer.py:208-247 on the MI355X path,
data-parallel over one process per GPU (new functionality - the reference never runs
multi-process).

  zero_grad -> forward (model(**batch)) -> backward -> [RCCL all-reduce of the flat fp32
  gradient buffer over xGMI] -> BertAdam.

bf16 compute with fp32 master weights replaces apex amp O2 (no loss scaling needed).
DP equivalence with a single-GPU run on the concatenated batch: every rank's loss is
re-weighted by n_local*world/n_global (token counts all-reduced on the device) before
backward, then gradients are averaged.  Ranks that see no target of a tail cluster still
contribute zeros: the flat buffer is always reduced as a whole."""
import os

import torch

from .. import hip, ops, streams
from .. import runtime as rt
from ..common.registrable import Registrable
from . import dp
from .optimizers import BertAdam, FlatParams, apply_no_grad


class TrainerBase(Registrable):
    pass


class Trainer:
    def __init__(self, model, optimizer_cfg=None, no_grad=(r'^resnet', r'^roberta'), device='cuda',
                 nan_check=False, bucket_mb=256, async_update=None, allreduce_dtype=None, data_parallel=None,
                 capture_after=None, shape_buckets=(128, 16)):
        """nan_check=True: the reference's host-synchronous NaN test (returns None for a skipped batch, no step graph);
        the default skips non-finite steps on the device instead (self.skip, skipped_steps()).

        capture_after: sightings of a batch shape before its step / encoder graphs are captured (default
        graphs.CAPTURE_AFTER = 2; fixed-shape runs pass 1).  shape_buckets = (article, caption): when graphs are in
        use, article ids are right-padded (pad id) to a multiple of `article` tokens and captions to a multiple of
        `caption`, faces to 4 and objects to 64 NaN rows - a handful of shapes instead of one per batch.  Padding is
        invisible to the loss: padded keys are masked in every attention, padded targets are ignore_index, the
        DynamicConv is causal.  None: leave batches as the iterator made them."""
        import torch.distributed as dist
        # data_parallel=False: a single-process trainer even though a process group exists (equivalence tests)
        self.dist = dist if (dist.is_available() and dist.is_initialized() and data_parallel is not False) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        # TELL_DP_SELFTEST=1: run every collective of the data-parallel path even in a 1-rank group (single-GPU check
        # of the RCCL calls, the streams they are issued on and their ordering)
        self.dp = self.world > 1 or (self.dist is not None and os.environ.get('TELL_DP_SELFTEST') == '1')
        self.rank = self.dist.get_rank() if self.dist else 0
        self.model = model.to(device)
        apply_no_grad(model, no_grad)
        if hasattr(model, 'reset_graphs'):           # requires_grad flags decide where the working weight copies live
            model.reset_graphs()
        self.flat = FlatParams(model.named_parameters(), device)
        cfg = dict(lr=1e-4, warmup=0.05, t_total=437600, schedule='warmup_linear', b1=0.9, b2=0.98, e=1e-6,
                   weight_decay=1e-5, max_grad_norm=0.1)                 # config.yaml:126-136
        cfg.update(optimizer_cfg or {})
        cfg.pop('type', None)
        self.optimizer = BertAdam(self.flat, **cfg)
        self.nan_check = nan_check
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self.batch_num_total = 0
        # gradients travel in bf16 when the model computes in bf16 (dp.all_reduce_flat_bf16), in fp32 in parity mode
        if allreduce_dtype is None:
            allreduce_dtype = torch.bfloat16 if (rt.compute_dtype() == torch.bfloat16 and
                                                 torch.device(device).type == 'cuda' and
                                                 os.environ.get('TELL_ALLREDUCE_FP32') != '1') else torch.float32
        self.allreduce_dtype, self._wire = allreduce_dtype, None
        if self.dp and self.world > 1:
            # RCCL's channel kernels will hold compute units while the next batch's encoders run: the 256x256 GEMMs
            # (csrc/gemm_q4.hip, one workgroup per CU) then take their tiles from per-XCD counters, so that a workgroup
            # whose CU is taken gets fewer tiles instead of finishing a static share late.  MEASURED on one GPU with N
            # CUs held by idle workgroups (bench.py --cu-hog N, profiles/r04_cu_hog.txt): configs[2] 1565 -> 1423 samples/s
            # with static tile lists for N = 8 / 16 / 32, -> 1498-1514 with the counters (1570 with no CU held).
            if 'TELL_Q4_DYNAMIC' not in os.environ:           # (an explicit choice at load time stands)
                hip.set_option('q4_dynamic', 1)
        if self.dp:
            # every rank draws its own dropout masks: the counter-hash seed is process-global with the same default on
            # all ranks; the rank is mixed in where the seed is READ (runtime.seed), not written into the global seed
            rt.set_rank(self.rank)
        if self.dp:                           # identical initial weights on every rank
            self.dist.broadcast(self.flat.flat, src=0)
            self.flat.refresh_shadow()
            rt.bump_weights_epoch()
        # TELL_ASYNC_UPDATE=1: gradient all-reduce + BertAdam + gradient zeroing run on their own stream and the model
        # waits for it right before its first trainable weight (rt.wait_weight_update).  Off by default: with the
        # encoders of the next batch already running on their two streams, a fourth / fifth stream of ours costs more
        # in hardware-queue sharing than the overlap returns (streams.py: 1149 vs 1134 vs 1061 samples/s for
        # 3 / 4 / 5 streams)
        if async_update is None:
            async_update = os.environ.get('TELL_ASYNC_UPDATE', '0') == '1'
        self.async_update = async_update and torch.device(device).type == 'cuda'
        self.update_stream = streams.get('update', device) if self.async_update else None
        self.flat.zero_grad()
        # gradient stores (ops.py): the first eager backward pass is observed, parameters with one dense whole-tensor
        # gradient write per step are then stored by their producer and left alone by BertAdam's zeroing.
        # TELL_GRAD_STORE=0: everything keeps zero + accumulate.
        self._stored_last = False                # the buffer holds gradients that the last update did not zero
        self._store = 'want' if (torch.device(device).type == 'cuda' and
                                 os.environ.get('TELL_GRAD_STORE', '1') != '0') else 'off'
        self.model.register_state_dict_pre_hook(lambda *a, **k: self.finish_update())
        # DP: the gradients of a decoder layer are final when backward leaves the layer (ops.grad_ready_marker); their
        # slice of the flat buffer is handed to RCCL right then (cast to the wire dtype on the backward stream, the
        # collective itself runs on RCCL's stream underneath the rest of backward).  Only what is left (embedder / tied
        # adaptive tables, 35 % of the bytes) is exchanged after backward.
        self._ranges, self._reduced, self._pending, self._in_backward = {}, [], [], False
        self._capturing = False
        # defer_update = True: a step stops after backward - gradients stay in flat.grad (they accumulate over calls
        # until the caller runs optimizer.step(zero_grad=True) or flat.zero_grad()); the captured step graph then ends
        # after backward as well.  Gradient accumulation over micro-batches, and how the parity tests read the
        # gradients a graph REPLAY produced.
        self.defer_update = False
        on_gpu = torch.device(device).type == 'cuda'
        # device-side NaN / Inf skip: [flag, count] read by the optimizer kernel (csrc/optim.hip) - the reference's
        # skip of a NaN batch (:225-227) and apex O2's overflow skip without a host synchronisation
        self.skip = torch.zeros(2, dtype=torch.int32, device=device) if on_gpu else None
        from .step_graph import StepGraph
        self.step_graph = StepGraph(self, capture_after) if on_gpu else None
        self.shape_buckets = tuple(shape_buckets) if shape_buckets else None
        if capture_after is not None and hasattr(model, 'set_capture_after'):
            model.set_capture_after(capture_after)                 # the encoder graphs follow the same policy
        self.bucketed_reduces = 0
        # dp_timing = []: every update then records three events on the update's stream - before the exchange is
        # handed to RCCL, after (casts + collective launches issued), and after the stream has waited for the last
        # bucket; dp_times() turns them into (allreduce_ms, exposed_allreduce_ms) per step
        self.dp_timing = None
        self._test_reduce_scale = None              # tests: emulate world_size 2 with identical ranks (x2 after a reduce)
        # The bucketed exchange needs Python inside backward, i.e. the eager schedule: it is used when the step graph is
        # off (TELL_STEP_GRAPH=0); with the graph the exchange follows backward and hides under the next batch's encoders
        from . import step_graph as _sg
        bucketed_default = '0' if (_sg.ENABLED and on_gpu) else '1'
        if self.dp and on_gpu and os.environ.get('TELL_DP_BUCKETED', bucketed_default) != '0':
            self._ranges = self._layer_ranges()
            if self._ranges:
                rt.set_grad_ready_callback(self._grad_ready)
                self.step_graph = None                   # the exchange is driven from Python inside backward

    def _layer_ranges(self):
        """{prefix: (lo, hi)} - the contiguous slice of the flat buffers that holds exactly the parameters whose name
        starts with 'decoder.layers.<i>.'."""
        fl, out = self.flat, {}
        ends = fl.offsets[1:] + [fl.total]
        for i in range(len(getattr(getattr(self.model, 'decoder', None), 'layers', []))):
            pre = 'decoder.layers.%d.' % i
            idx = [k for k, n in enumerate(fl.names) if n.startswith(pre)]
            if not idx or idx != list(range(idx[0], idx[-1] + 1)):
                return {}                            # not contiguous (tied across layers): keep the single exchange
            out[pre] = (fl.offsets[idx[0]], ends[idx[-1]])
        return out

    def _grad_ready(self, tag):
        """Called from the autograd thread (current stream = the stream backward runs on)."""
        rng_ = self._ranges.get(tag)
        if rng_ is None or not self._in_backward or self._capturing:
            return
        side = ops.flush_wgrad_stream()              # only with the opt-in weight-gradient stream
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        with hip.bound_stream():
            self._start_reduce(*rng_)
        self._reduced.append(rng_)
        self.bucketed_reduces += 1

    def dp_times(self):
        """-> [(allreduce_ms, exposed_allreduce_ms)] of the updates recorded since dp_timing was set (call after a device
        synchronisation).  allreduce_ms: from handing the gradient to RCCL until the update stream holds the reduced
        gradient; exposed: the part of it the update stream spent waiting (nothing of its own left to run) - the
        other streams (the next batch's encoders) keep the GPU busy meanwhile."""
        out = [(e[0].elapsed_time(e[2]), e[1].elapsed_time(e[2])) for e in (self.dp_timing or [])]
        if self.dp_timing is not None:
            self.dp_timing = []
        return out

    def finish_update(self):
        """Make the current stream wait for an in-flight weight update (checkpointing, evaluation, ...)."""
        rt.wait_weight_update()

    def train_one_batch(self, batch, next_batch=None):
        """callback_apex_trainer.py:208-247 for one batch; returns the (detached) loss tensor.

        next_batch: the batch that will be trained next, if the caller already has it (a data loader always
        does).  Its frozen encoders are launched on their own streams BEFORE this batch's decoder work is
        issued, so ResNet-152 / RoBERTa-large of step N+1 fill the GPU underneath the latency-bound decoder
        forward, backward and optimizer of step N.  Numerics are unchanged: the encoders read no trainable
        weight, and BatchNorm running statistics are still updated in batch order on the encoder stream."""
        with hip.bound_stream():
            return self._train_one_batch(batch, next_batch)

    def _encoded_for(self, batch):
        pre, self._prefetched = getattr(self, '_prefetched', None), None
        if pre is not None and pre[0] is batch.get('image') and not pre[1].stale():
            return pre[1]           # (stale: other encode() / generate() calls in between reused the graph's buffers)
        return None

    def _bucketed(self, batch):
        """The batch padded to the shape buckets (see __init__); a batch that already fits is returned as it is."""
        from . import step_graph as _sg
        from .. import graphs
        if self.shape_buckets is None or not (_sg.ENABLED and graphs.ENABLED) or self.nan_check:
            return batch
        idx = getattr(self.model, 'index', 'roberta')
        pad = int(getattr(self.model, 'padding_idx', 1))
        out = None

        def grow(t, dim, n, value):
            shape = list(t.shape)
            shape[dim] = n - t.shape[dim]
            return torch.cat([t, t.new_full(shape, value)], dim=dim)
        # a caption of T + 1 tokens makes T decoder positions: T is what gets bucketed
        for key, mult, extra in (('context', self.shape_buckets[0], 0), ('caption', self.shape_buckets[1], 1)):
            field = batch.get(key)
            ids = field.get(idx) if isinstance(field, dict) else None
            if not (torch.is_tensor(ids) and ids.is_cuda and ids.dim() == 2):
                continue
            n = -(-(ids.shape[1] - extra) // mult) * mult + extra
            if n != ids.shape[1]:
                out = dict(batch) if out is None else out
                f = dict(field)
                f[idx] = grow(ids, 1, n, pad)
                m = f.get(idx + '_copy_masks')
                if torch.is_tensor(m) and m.dim() == 2 and m.shape[1] == ids.shape[1]:
                    f[idx + '_copy_masks'] = grow(m, 1, n, -1)
                out[key] = f
        for key, rows in (('face_embeds', 4), ('obj_embeds', 64)):
            t = batch.get(key)
            if torch.is_tensor(t) and t.is_cuda and t.dim() == 3 and t.shape[2] > 0 and t.shape[1] < rows:
                out = dict(batch) if out is None else out
                out[key] = grow(t, 1, rows, float('nan'))
        return batch if out is None else out

    @staticmethod
    def _store_safe(batch):
        """Store mode rests on every marked tensor being rewritten WHOLE by this step's backward pass.  The one
        data-dependent exception: a batch in which no sample has a face / an object hands the model an EMPTY context
        ([B,1,0], collate's empty field), for which the K / V projections of that context are skipped (blocks.py
        kv_project_all, modules/attention.py project_kv; multi_head.py:349-374) - rows [E,3E) of its in_proj_weight
        then get no gradient write, and a stored buffer would still hold the previous step's rows.  Such a step runs
        with the zero + accumulate convention (shape test on the host: no synchronisation)."""
        for key in ('face_embeds', 'obj_embeds'):
            t = batch.get(key) if isinstance(batch, dict) else None
            if torch.is_tensor(t) and t.dim() == 3 and (t.shape[1] == 0 or t.shape[2] == 0):
                return False
        return True

    def _grad_store_begin(self, batch=None):
        """Select this step's gradient convention.  Store mode needs an update after every backward pass (micro-batch
        accumulation - defer_update - adds into the buffer) and a batch whose backward pass writes every marked tensor
        (_store_safe); leaving it, the stored gradients of the last step are still in the buffer and have to go.  While
        the buffer holds accumulated gradients that no update has consumed yet (a defer_update pass came before), the
        pass that ends the accumulation must ADD to them too: store mode stays off until an update / zero_grad has
        cleared the buffer (flat.accum_pending)."""
        want = (self._store == 'ready' and not self.defer_update and not self.flat.accum_pending and
                (batch is None or self._store_safe(batch)))
        if self._stored_last and not want:       # (this trainer's own history: the switch in ops.py is process-wide)
            pending = self.flat.accum_pending
            self.flat.zero_grad()
            self.flat.accum_pending = pending
        self._stored_last = want
        ops.grad_store_mode(want)
        if self.defer_update:
            self.flat.accum_pending = True

    def _train_one_batch(self, batch, next_batch=None):
        try:
            return self._train_one_batch_body(batch, next_batch)
        finally:
            ops.grad_store_mode(False)           # outside a trainer step every backward pass accumulates

    def _train_one_batch_body(self, batch, next_batch=None):
        if not self.model.training:              # (recursing through ~650 modules costs 2.5 ms of host time)
            self.model.train()                   # (:214 zero_grad: done right after the previous update)
        batch = self._bucketed(batch)
        self._grad_store_begin(batch)
        if next_batch is not None:
            next_batch = self._bucketed(next_batch)
        enc = None
        if hasattr(self.model, 'encode') and torch.is_tensor(batch.get('image')) and batch['image'].is_cuda:
            enc = self._encoded_for(batch)
            if enc is None:
                enc = self.model.encode(batch['context'], batch['image'])
            if next_batch is not None:
                self._prefetched = (next_batch['image'],
                                    self.model.encode(next_batch['context'], next_batch['image'], ahead=True))
        loss = None
        if enc is not None and self.step_graph is not None and not self.nan_check:
            loss = self.step_graph.run(batch, enc, self._eager_step)      # decoder fwd + loss + bwd (+ update) replayed
        if loss is None:
            loss = self._eager_step(batch, enc)
        if loss is not None:
            self.batch_num_total += 1
        return loss

    def _flag_loss(self, loss):
        """skip[0] = the loss is NaN / Inf (:225-227 skips such a batch; here the optimizer kernel does, no host sync)."""
        if loss.is_cuda:
            hip.call('tell_loss_flag', loss.detach().reshape(1).float(), self.skip)

    def _backward(self, scaled):
        self._in_backward = True
        # the weight-norm chain rule of all GehringLinears runs as ONE launch after the pass - unless gradient buckets
        # leave during backward (a layer's gradients have to be final at its marker then)
        ops.wn_defer(not self._ranges)
        ops.wgrad_group_defer(not self._ranges)                          # ... and so do the weight-gradient GEMMs
        ops.finish_defer(not self._ranges)                               # ... and the small column-sum finishers
        observe = self._store == 'want' and not self._capturing
        if observe:
            ops.grad_store_observe(True)
        try:
            scaled.backward()                                            # :229-231
        except BaseException:
            ops.wn_drop()
            ops.wgrad_group_drop()
            ops.finish_drop()
            if observe:
                ops.grad_store_observe(False)
            raise
        finally:
            self._in_backward = False
            ops.wn_defer(False)
            ops.wgrad_group_defer(False)
            ops.finish_defer(False)
        ops.join_wgrad_stream()                                          # weight-gradient side stream (ops.py)
        ops.wgrad_group_flush()
        ops.wn_flush()
        ops.finish_flush()
        if observe:
            # the gradients in the buffer are complete and were accumulated onto zeros: the convention can change right
            # here - this step's update already leaves the marked tensors alone, the next pass stores over them.  (Every
            # graph is captured after this point: the convention a signature is captured with follows from defer_update,
            # which is part of the signature.)
            self.flat.set_grad_store(ops.grad_store_observe(False))
            self._store = 'ready'
            self._stored_last = not self.defer_update
            ops.grad_store_mode(self._stored_last)

    def skipped_steps(self):
        """Number of optimisation steps the device-side NaN / Inf check turned into no-ops so far (one host sync)."""
        return int(self.skip[1].item()) if self.skip is not None else 0

    def _eager_step(self, batch, enc=None):
        extra = {'encoded': enc} if enc is not None else {}
        out = self.model(**batch, **extra)                               # :220 / :194
        loss = out['loss']
        self._flag_loss(loss)
        n_local = None
        if self.dp and self._ranges:
            # bucketed exchange during backward: the per-rank weight has to be in the gradients before the first
            # bucket leaves, i.e. on the loss
            scaled = loss * dp.loss_weight(out['sample_size'].to(torch.float32).reshape(1), self.dist,
                                           self.world).reshape(())
        else:
            scaled = loss
            n_local = out['sample_size'] if self.dp else None
        if self.nan_check:                                               # :225-227 with the reference's host sync
            bad = (~torch.isfinite(loss.detach())).to(torch.float32)
            if self.dp:
                self.dist.all_reduce(bad)
            if bad.item() > 0:
                self.flat.zero_grad()
                return None
        self._backward(scaled)
        if self.async_update:
            main = torch.cuda.current_stream()
            self.update_stream.wait_stream(main)
            with torch.cuda.stream(self.update_stream), hip.bound_stream():
                self._update(n_local)
                done = torch.cuda.Event()
                done.record(self.update_stream)
            rt.set_pending_update(done)
        else:
            self._update(n_local)
        return loss.detach()

    def _update(self, n_local=None):
        """Gradient exchange (data parallel) + BertAdam + gradient zeroing.  n_local: this rank's token count when
        the loss was NOT weighted before backward (graph replay, unbucketed eager): the weight n_local * world /
        n_global is then applied to the gradient on its way to the wire."""
        if self.defer_update:
            return
        if self.dp:
            scale = None
            ev = None
            if self.dp_timing is not None and self.flat.grad.is_cuda:      # bench / diagnosis: events, read after a sync
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
            if n_local is not None:
                scale = dp.loss_weight(n_local.to(torch.float32).reshape(1), self.dist, self.world)
            if self.skip is not None and self.skip.is_cuda:
                self.dist.all_reduce(self.skip[:1], op=self.dist.ReduceOp.MAX)        # collective NaN skip
            done, self._reduced = sorted(self._reduced), []
            lo = 0
            for a, b in done + [(self.flat.total, self.flat.total)]:     # everything not exchanged during backward
                if a > lo:
                    self._start_reduce(lo, a, scale)
                lo = max(lo, b)
            if ev is not None:
                ev[1].record()
            wire = self._finish_reduces()
            if ev is not None:
                ev[2].record()
                self.dp_timing.append(ev)
            if wire is not None:                       # bf16 on the wire: BertAdam reads the reduced buffer as it is
                self.optimizer.step(grad_scale=1.0 / self.world, zero_grad=True, skip=self.skip, grad_wire=wire)
                return
        self.optimizer.step(grad_scale=1.0 / self.world, zero_grad=True, skip=self.skip)  # :238 (+ :214 of the next batch)

    def _cast(self, src, dst):
        hip.call('tell_cast', src, hip.dt(src), dst, hip.dt(dst), src.numel())

    def _start_reduce(self, lo, hi, scale=None):
        """Hand flat.grad[lo:hi] to RCCL (asynchronously; bf16 on the wire in bf16 mode, see dp.all_reduce_flat_bf16).
        scale: optional device scalar multiplied in on the way (the rank's loss weight)."""
        grad = self.flat.grad[lo:hi]
        if self.allreduce_dtype == torch.bfloat16:
            if self._wire is None:
                self._wire = torch.empty(self.flat.total, dtype=torch.bfloat16, device=self.flat.grad.device)
            buf, step = self._wire[lo:hi], 2 * self.bucket_elems
            hip.call('tell_scale_cast', grad, buf, hip.BF16, grad.numel(), scale)
        else:
            buf, step = grad, self.bucket_elems
            if scale is not None:
                if grad.is_cuda:
                    hip.call('tell_scale_cast', grad, grad, hip.F32, grad.numel(), scale)
                else:
                    grad.mul_(scale)
        handles = [self.dist.all_reduce(buf[s:s + step], async_op=True) for s in range(0, buf.numel(), step)]
        self._pending.append((lo, hi, handles))

    def _finish_reduces(self):
        """The current stream waits for every exchange in flight.  -> the bf16 wire buffer when it now holds the WHOLE
        reduced gradient and the optimizer kernel can read it directly (CUDA, bf16 on the wire, one exchange covering
        all of flat.grad), else None after widening the bf16 results back into flat.grad."""
        pending, self._pending = self._pending, []
        for lo, hi, handles in pending:
            for h in handles:
                h.wait()
        direct = (self.allreduce_dtype == torch.bfloat16 and self._test_reduce_scale is None and
                  self.flat.grad.is_cuda and len(pending) == 1 and pending[0][0] == 0 and
                  pending[0][1] == self.flat.total and hasattr(self.optimizer, 'launch'))
        if direct:
            return self._wire
        for lo, hi, handles in pending:
            grad = self.flat.grad[lo:hi]
            if self.allreduce_dtype == torch.bfloat16:
                self._cast(self._wire[lo:hi], grad)
            if self._test_reduce_scale is not None:
                grad.mul_(self._test_reduce_scale)
        return None


@TrainerBase.register('callback_apex')
class CallbackApexTrainer(Trainer):
    """Registration name of the reference trainer (callback_apex_trainer.py:51).  The AllenNLP
    callback machinery (checkpoint / tensorboard / validation callbacks) is out of scope;
    `apex_opt_level` / `keep_batchnorm_fp32` are accepted and ignored (bf16 + fp32 masters)."""

    def __init__(self, model, optimizer=None, no_grad=(r'^resnet', r'^roberta'), apex_opt_level=None,
                 keep_batchnorm_fp32=None, num_epochs=1, shuffle=True, cuda_device=0, callbacks=None, **kw):
        super().__init__(model, optimizer, no_grad, device='cuda' if cuda_device is not None else 'cpu')
        self.num_epochs = num_epochs

    def train(self, batches):
        losses = []
        for batch in batches:
            losses.append(self.train_one_batch(batch))
        return losses
