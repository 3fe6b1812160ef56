"""Epilogue cost in isolation: the fc1 output shape with K = 64 (one K tile), act 0 / 1 / 2, bias on/off."""
import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import ops
M, N, K = 8192, 4096, 64
a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
bias = torch.randn(N, device='cuda')
out = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
for act, bm in ((0, 0), (0, 1), (1, 1), (2, 1), (2, 0)):
    f = lambda: ops.gemm(a, b, out=out, bias=bias if bm else None, bias_mode=bm, act=act)
    for _ in range(5): f()
    torch.cuda.synchronize()
    ts = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 30)
    print('act %d bias %d: %6.1f us' % (act, bm, sorted(ts)[1]), flush=True)
