import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import tell_amd
from tell_amd import ops
M, N, K = 8192, 4096, 1024
a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
bias = torch.randn(N, device='cuda'); out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
junk = torch.empty(1 << 28, device='cuda', dtype=torch.bfloat16)      # 512 MB: evicts the 256 MB Infinity Cache
def t(fn, flush, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if flush: junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot * 1e3 / n
for name, fn in [('plain', lambda: ops.gemm(a, b, out=out)), ('bias', lambda: ops.gemm(a, b, out=out, bias=bias, bias_mode=1)),
                 ('bias+gelu', lambda: ops.gemm(a, b, out=out, bias=bias, bias_mode=1, act=2))]:
    print('  %-10s warm %6.1f us   cold (cache flushed) %6.1f us' % (name, t(fn, False), t(fn, True)), flush=True)
'''
for v in ['0', '1']:
    print('TELL_GEMM_TILE=%s (%s)' % (v, '256x256 for this shape' if v == '0' else '128x128'), flush=True)
    subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, TELL_GEMM_TILE=v))
