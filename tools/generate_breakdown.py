#!/usr/bin/env python
"""Where a batch of captions spends its time (configs[4]): encoders, K/V projection, stepper set-up, decode loop.
usage (GPU box): python tools/generate_breakdown.py [beam]"""
import sys
import time

import torch

sys.path.insert(0, '.')
import tell_amd  # noqa: E402
from tell_amd.build import build_model  # noqa: E402
from tell_amd.data import synthetic_batch  # noqa: E402

beam = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tell_amd.hip.require_gpu()
tell_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = build_model('faces_objects').to('cuda').eval()
batches = [synthetic_batch(32, 512, 33, True, seed=4321 + 97 * i, device='cuda') for i in range(2)]


def clone(b):
    return {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items() if k != 'metadata'}


def t():
    torch.cuda.synchronize()
    return time.perf_counter()


with torch.no_grad(), tell_amd.hip.bound_stream():
    for rep in range(3):
        b = clone(batches[rep % 2])
        t0 = t()
        cap_ids, _, contexts = model._forward(**b)
        t1 = t()
        kv = model.decoder.project_contexts(contexts)
        t2 = t()
        if beam > 1:
            out = model._generate_beam(cap_ids, contexts, beam)
        else:
            out = model._generate_cached(cap_ids, contexts)
        t3 = t()
        print('rep %d: encoders %.1f ms, K/V projection (standalone) %.1f ms, generate (incl. its own K/V projection, stepper '
              'set-up, %d steps) %.1f ms' % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), out[1].shape[1] - 1, 1e3 * (t3 - t2)))
    # host cost of the loop alone: issue without waiting
    b = clone(batches[0])
    cap_ids, _, contexts = model._forward(**b)
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    out = model._generate_beam(cap_ids, contexts, beam) if beam > 1 else model._generate_cached(cap_ids, contexts)
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    h2 = time.perf_counter()
    print('generate: host returned after %.1f ms, device done after %.1f ms' % (1e3 * (h1 - h0), 1e3 * (h2 - h0)))
