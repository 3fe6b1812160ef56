"""Optimizer `bert_adam` (expt/nytimes/9_transformer_objects/config.yaml:126-149) as one
multi-tensor HIP kernel over flat fp32 buffers.  Update rule restated from
pytorch_pretrained_bert.BertAdam (third-party, absent here; parity unpinned):
per-tensor gradient-norm clipping, no bias correction, decoupled-style weight decay added
to the update, `warmup_linear` schedule evaluated at the step count BEFORE the increment."""
import re

import torch

from .. import hip
from .. import runtime as rt


def warmup_linear(progress, warmup):
    """pytorch_pretrained_bert WarmupLinearSchedule.get_lr_"""
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


class FlatParams:
    """Trainable parameters re-homed into one flat fp32 buffer (+ matching grad / m / v
    buffers).  Every tensor starts on a CHUNK boundary so a chunk belongs to one tensor
    (per-tensor norms without atomics); `p.data` and `p.grad` become views."""

    def __init__(self, named_params, device):
        chunk = hip.lib().tell_opt_chunk()
        self.names, self.params, offs = [], [], []
        total = 0
        seen = set()
        for name, p in named_params:
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            self.names.append(name)
            self.params.append(p)
            offs.append(total)
            total += (p.numel() + chunk - 1) // chunk * chunk
        self.total, self.chunk, self.offsets = total, chunk, offs
        self.n_chunks = total // chunk
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.m = torch.zeros(total, dtype=torch.float32, device=device)
        self.v = torch.zeros(total, dtype=torch.float32, device=device)
        chunk_tensor = torch.empty(self.n_chunks, dtype=torch.int32)
        begins = []
        for i, (p, o) in enumerate(zip(self.params, offs)):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = self.flat[o:o + n].view(p.shape)
            g = self.grad[o:o + n].view(p.shape)
            p.grad = g
            p._tell_grad = g
            c0, c1 = o // chunk, (o + (n + chunk - 1) // chunk * chunk) // chunk
            chunk_tensor[c0:c1] = i
            begins.append(c0)
        begins.append(self.n_chunks)
        self.chunk_tensor = chunk_tensor.to(device)
        self.chunk_begin = torch.tensor(begins, dtype=torch.int64, device=device)
        # bf16 working copy of every parameter, rewritten by the optimizer kernel itself (ops.weight reads it)
        self.shadow = None
        if torch.device(device).type == 'cuda':
            self.shadow = torch.empty(total, dtype=torch.bfloat16, device=device)
            for p, o in zip(self.params, offs):
                p._tell_shadow = self.shadow[o:o + p.numel()].view(p.shape)
            self.refresh_shadow()
        # tensors whose gradient the optimizer's zeroing leaves alone (their single dense producer stores over it:
        # ops.py 'Gradient stores'); all zero until the trainer has observed a step (set_grad_store)
        self.keep_grad = torch.zeros(max(len(self.params), 1), dtype=torch.int32, device=device)
        self.stored_numel = 0
        # True while the buffer holds gradients of defer_update passes that no update has consumed (trainer.py
        # _grad_store_begin: the pass that ends the accumulation must add to them, not store over part of them)
        self.accum_pending = False
        self.partial = torch.empty(max(self.n_chunks, 1), dtype=torch.float32, device=device)
        self.norms = torch.zeros(len(self.params), dtype=torch.float32, device=device)
        rt.bump_weights_epoch()

    def refresh_shadow(self):
        """Re-derive the bf16 working copy from the fp32 masters (after anything but the optimizer kernel wrote the
        flat buffer: construction, the initial DP broadcast, a checkpoint load into `flat`)."""
        if self.shadow is not None:
            hip.call('tell_cast', self.flat, hip.F32, self.shadow, hip.BF16, self.total)
            for p in self.params:
                p._tell_shadow_version = p._version

    def zero_grad(self):
        hip.call('tell_fill_f32', self.grad, self.total, 0.0)
        self.accum_pending = False

    def set_grad_store(self, seen):
        """seen: ops.grad_store_observe(False) of one whole backward pass - {id(p): (p, [(r0, r1) | None, ...])}, a None
        entry being an accumulating writer.  A parameter whose gradient rows were written exactly once each, by dense
        products only, is marked `_tell_grad_store` (and kept out of the optimizer's zeroing).  -> elements marked."""
        flags = []
        for p in self.params:
            e = seen.get(id(p))
            ok = e is not None and e[0] is p and len(e[1]) > 0 and all(w is not None for w in e[1])
            if ok:
                pos = 0
                for r0, r1 in sorted(e[1]):
                    ok = ok and r0 == pos
                    pos = r1
                ok = ok and pos == p.shape[0]
            p._tell_grad_store = bool(ok)
            flags.append(int(ok))
        # the (g, v) pair of a GehringLinear is written by ONE launch with ONE store bit (ops._wn_backward): mark jointly
        index = {n: i for i, n in enumerate(self.names)}
        for n, ia in index.items():
            if n.endswith('.weight_g') and n[:-1] + 'v' in index:
                ib = index[n[:-1] + 'v']
                if flags[ia] != flags[ib]:
                    flags[ia] = flags[ib] = 0
                    self.params[ia]._tell_grad_store = self.params[ib]._tell_grad_store = False
        if flags:
            self.keep_grad.copy_(torch.tensor(flags, dtype=torch.int32))
        self.stored_numel = sum(p.numel() for p, f in zip(self.params, flags) if f)
        return self.stored_numel

    def numel(self):
        return sum(p.numel() for p in self.params)


class BertAdam:
    def __init__(self, flat, lr=1e-4, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999,
                 e=1e-6, weight_decay=0.01, max_grad_norm=1.0, parameter_groups=None):
        assert schedule == 'warmup_linear'
        self.flat = flat
        self.lr, self.warmup, self.t_total = lr, warmup, t_total
        self.b1, self.b2, self.e, self.weight_decay, self.max_grad_norm = b1, b2, e, weight_decay, max_grad_norm
        self.step_count = 0         # optimisation steps ISSUED by the host (incl. steps the device skipped)
        dev = flat.flat.device
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        # updates APPLIED: advanced by the update kernel itself, only when the step is not skipped for a non-finite
        # loss / gradient; the learning rate of a step is computed from it on the device (csrc/optim.hip)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.parameter_groups = parameter_groups    # all groups of the configs carry `{}` overrides

    def applied_steps(self):
        """Updates applied so far = issued steps minus the skipped ones (one host sync)."""
        return int(self.step_dev.item())

    def current_lr(self, step=None):
        """Learning rate of the step that follows `step` applied updates (default: the host's count of issued steps;
        pass applied_steps() for what the device will actually use after skipped batches)."""
        step = self.step_count if step is None else step
        if self.t_total != -1:
            return self.lr * warmup_linear(step / self.t_total, self.warmup)
        return self.lr

    def step(self, grad_scale=1.0, zero_grad=False, skip=None, grad_wire=None):
        """zero_grad: also clear the gradient buffer (fused into the update pass).  skip: device int32[2] flag -
        a non-zero skip[0] (non-finite loss / gradient) turns the step into gradient zeroing only."""
        self.prepare()
        self.launch(grad_scale, zero_grad, skip, grad_wire)
        self.advance()

    # the three parts of step(), separable so that the kernel launches can live in a captured graph
    def prepare(self):
        """Host part before the kernels: nothing any more - the schedule lives on the device (step_dev)."""

    def launch(self, grad_scale=1.0, zero_grad=False, skip=None, grad_wire=None):
        """grad_wire: optional bf16 copy of the whole flat gradient (the data-parallel exchange's wire buffer after the
        all-reduce): the kernels read it in place of flat.grad - no pass that widens it back first."""
        f = self.flat
        from .. import ops
        keep = f.keep_grad if (zero_grad and ops.grad_store_on()) else None
        if zero_grad:
            f.accum_pending = False
        hip.call('tell_bertadam_step2', f.flat, f.grad, f.m, f.v, f.chunk_tensor, f.chunk_begin, f.n_chunks,
                 len(f.params), f.partial, f.norms, self.lr_dev, self.b1, self.b2, self.e, self.weight_decay,
                 self.max_grad_norm, float(grad_scale), f.shadow, int(zero_grad), skip, grad_wire,
                 self.step_dev, float(self.lr), float(self.warmup), float(self.t_total), keep)

    def advance(self):
        """Host part after the kernels.  The host cannot see a device-side skip without a synchronisation, so it bumps
        the weights epoch regardless: after a skipped step that costs one needless rebuild of the cached working
        weights in the eager schedule (the step graph rebuilds them inside the graph anyway) - never a stale one."""
        self.step_count += 1
        rt.bump_weights_epoch()

    def state_dict(self):
        return {'step': self.applied_steps(), 'm': self.flat.m, 'v': self.flat.v}

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        self.step_dev.fill_(int(sd['step']))
        self.flat.m.copy_(sd['m'])
        self.flat.v.copy_(sd['v'])


def apply_no_grad(model, patterns):
    """trainer `no_grad` regex list (config.yaml:150-152)."""
    for name, p in model.named_parameters():
        if any(re.search(rx, name) for rx in patterns):
            p.requires_grad_(False)
