"""The side streams of the step schedule.

ROCm multiplexes HIP streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and streams that
share a queue run FIFO.  The production schedule therefore uses THREE streams - the default stream (decoder forward /
backward, weight gradients, optimizer), `roberta` and `resnet` (the frozen encoders of the next batch, replayed as
hipGraphs) - which leaves the fourth queue to an RCCL communicator.  Measured on MI355X (samples/s, 1 GPU, same box):

  3 streams (default)                              1149
  + `update` stream  (TELL_ASYNC_UPDATE=1)         1134
  + `wgrad` stream   (TELL_WGRAD_STREAM=1) too     1061   <- the weight-gradient GEMMs share RoBERTa's hardware queue
  1-rank RCCL group, GPU_MAX_HW_QUEUES=4 / 3       1120 / 885
  GPU_MAX_HW_QUEUES = 3 / 5 / 6 / 8 (no RCCL)       933 / 660 / ~800 / ~780
  low-priority encoder streams / high-priority update stream (hipStreamCreateWithPriority)   640 / 640  (five streams)
  high-priority RoBERTa stream / low-priority ResNet stream with three streams              -2 % / 0 %
  ResNet replay on the main / on RoBERTa's stream  943 / 957 (own stream: 1170)

(Before the encoders were prefetched and graph-replayed, the two extra streams were a gain - they are kept as opt-in
paths and covered by tests/test_gpu_train.py.)  The creation order of the streams does not change which of them share a
queue (four orders: 1065-1069 with five streams).  Roles outside ROLES are created on demand."""
import torch

import os

ROLES = tuple(os.environ.get('TELL_STREAM_ORDER', 'resnet,roberta').split(','))   # wgrad / update: created on demand (opt-in)
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
