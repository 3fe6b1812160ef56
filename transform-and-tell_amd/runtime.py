"""Process-wide runtime state of the MI355X path: compute dtype, dropout RNG
counters, weight-cache epoch.  PyTorch is used only as the device-memory /
stream / autograd-graph plumbing around the HIP kernels."""
import torch

_state = {
    'dtype': torch.bfloat16,   # activations / working weights; torch.float32 = exact parity mode
    'seed': 0x5EED,
    'salt': 0,
    'rank': 0,                 # data-parallel rank: folded into seed() so that every rank draws its own masks
    'epoch': 0,                # bumped whenever master weights change (optimizer step / load)
}


def compute_dtype():
    return _state['dtype']


def set_compute_dtype(dtype):
    assert dtype in (torch.float32, torch.bfloat16)
    _state['dtype'] = dtype
    bump_weights_epoch()


def manual_seed(seed, salt=0):
    """Seed of the counter-based dropout RNG (csrc/common.h tell_keep_field)."""
    _state['seed'] = int(seed) & 0xFFFFFFFF
    _state['salt'] = int(salt) & 0xFFFFFFFF


def next_salt():
    """A fresh salt per dropout site per forward call; backward re-uses it."""
    _state['salt'] = (_state['salt'] + 1) & 0xFFFFFFFF
    return _state['salt']


def set_rank(rank):
    """Data-parallel rank of this process.  seed() mixes it in at use time, so manual_seed(s) stays idempotent (the
    same call on every rank, any number of times, any number of Trainers) and ranks still never share a mask."""
    _state['rank'] = int(rank)


def seed():
    return (_state['seed'] + 0x9E3779B1 * _state['rank']) & 0xFFFFFFFF


def weights_epoch():
    return _state['epoch']


def bump_weights_epoch():
    _state['epoch'] += 1


# --------------------------------------------------------------------------- #
# asynchronous weight update
# --------------------------------------------------------------------------- #
# The trainer runs gradient all-reduce + optimizer + gradient zeroing on its own stream so that they overlap
# the NEXT step's frozen encoders (which read no trainable weight).  Whoever is about to read a trainable
# weight on another stream calls wait_weight_update() first.
_pending_update = [None]


_grad_ready = [None]


def set_grad_ready_callback(fn):
    """Data-parallel trainer: fn(tag) is called from the backward pass when every gradient of the parameters behind
    `ops.grad_ready_marker(x, tag)` has been issued (they can be all-reduced while the rest of backward runs)."""
    _grad_ready[0] = fn


def grad_ready_callback():
    return _grad_ready[0]


def set_pending_update(event):
    _pending_update[0] = event


def wait_weight_update():
    ev = _pending_update[0]
    if ev is not None:
        _pending_update[0] = None
        torch.cuda.current_stream().wait_event(ev)
