import sys, os, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import hip
L, n = 25, 16384 * 1024
H = torch.randn(L, n, device='cuda').bfloat16(); d = torch.randn(n, device='cuda').bfloat16()
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
for nb in (256, 512, 1024, 2048, 4096):
    partial = torch.empty(nb, L, device='cuda')
    t = timed(lambda: hip.call('tell_mix_bwd', H, d, L, n, partial, nb, hip.BF16))
    print('nb %4d: %.1f us  %.2f TB/s' % (nb, t, (L + 1) * n * 2 / t * 1e-6))
