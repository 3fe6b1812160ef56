import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mb in (16, 67, 134, 512, 2048):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device='cuda'); y = torch.empty_like(x)
    us = t(lambda: x.zero_()); print('fill %5d MB %8.1f us %6.2f TB/s' % (mb, us, mb / us))
    us = t(lambda: y.copy_(x)); print('copy %5d MB %8.1f us %6.2f TB/s (r+w)' % (mb, us, 2 * mb / us))
    us = t(lambda: x.sum()); print('read %5d MB %8.1f us %6.2f TB/s' % (mb, us, mb / us))
