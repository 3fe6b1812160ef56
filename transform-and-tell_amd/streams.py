"""The side streams of the step schedule, created in ONE canonical order.

ROCm multiplexes HIP streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in
creation order, and the command processor runs up to 4 active queues per process concurrently; beyond that it
time-slices them.  Measured on MI355X with this schedule (samples/s, 1 GPU): GPU_MAX_HW_QUEUES = 1: 778,
2: 835, 3: 1073, 4: 1125, 8: 779, 16: 760 - and 940 at the default 4 once an RCCL communicator exists (its
queue is a 5th), back to 1085-1093 with GPU_MAX_HW_QUEUES=3.  Hence: every role gets its stream from this one
registry in a fixed order, and multi-GPU launchers set GPU_MAX_HW_QUEUES=3 before the first HIP call
(bench.py does).
Stream priorities are not an option for the same reason: hipStreamCreateWithPriority streams get hardware queues of
their own (low-priority encoder streams: 1060 -> 640 samples/s, high-priority wgrad/update streams: -> 640).  Folding
roles together loses too: update on the wgrad stream -1 %, both encoders on one stream -11 %.  The creation order of the
streams does not change which of them share a hardware queue (four orders: 1065-1069 samples/s), and GPU_MAX_HW_QUEUES = 5 / 6
are as bad as 8 (780-830)."""
import torch

import os

ROLES = tuple(os.environ.get('TELL_STREAM_ORDER', 'resnet,roberta,wgrad,update').split(','))
_streams = {}


def get(role, device=None):
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (device.index, role)
    s = _streams.get(key)
    if s is None:
        warm(device)
        s = _streams.get(key)
        if s is None:                       # a role outside the canonical list
            s = _streams[key] = torch.cuda.Stream(device=device)
    return s


def warm(device=None):
    """Create all canonical streams of `device` now (idempotent)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    for role in ROLES:
        if (device.index, role) not in _streams:
            _streams[(device.index, role)] = torch.cuda.Stream(device=device)
