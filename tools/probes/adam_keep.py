"""BertAdam update pass with and without keep_grad (tell_bertadam_step2): same flat buffers, interleaved timing.
200 M parameters in 64 tensors, 70 % of the elements in kept tensors."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip
from tell_amd.training.optimizers import FlatParams, BertAdam
hip.require_gpu()
ps = [torch.nn.Parameter(torch.randn(3 * 1024 * 1024 + 64 * i)) for i in range(64)]
flat = FlatParams([('p%d' % i, p) for i, p in enumerate(ps)], 'cuda')
opt = BertAdam(flat, lr=1e-4)
flat.grad.normal_()
flags = torch.tensor([1 if i % 10 < 7 else 0 for i in range(64)], dtype=torch.int32)
flat.keep_grad.copy_(flags)
f = flat


def run(keep, zero=1):
    hip.call('tell_bertadam_step2', f.flat, f.grad, f.m, f.v, f.chunk_tensor, f.chunk_begin, f.n_chunks, len(f.params),
             f.partial, f.norms, opt.lr_dev, 0.9, 0.999, 1e-6, 0.01, 1.0, 1.0, f.shadow, zero, None, None, opt.step_dev,
             1e-4, -1.0, -1.0, f.keep_grad if keep else None)


res = {False: [], True: []}
for r in range(9):
    for keep in (False, True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(keep)
        e1.record(); torch.cuda.synchronize()
        if r >= 2:
            res[keep].append(e0.elapsed_time(e1) / 5)
n = flat.total
for keep in (False, True):
    t = statistics.median(res[keep])
    print('keep_grad %-5s  norm pass + update %.3f ms for %.0f M parameters' % (keep, t, n / 1e6))
