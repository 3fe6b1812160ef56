"""GEMM tile-variant study: one subprocess per TELL_GEMM_TILE value (the dispatch reads it once)."""
import os, subprocess, sys
CHILD = r'''
import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import ops
shapes = [(8192, 3072, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (8192, 3072, 1024), (8192, 1024, 1024), (4096, 4096, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20; e0.record()
    for _ in range(n): ops.gemm(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print('  M%-5d N%-5d K%-5d %8.1f us %7.1f TF/s' % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
'''
for v in sys.argv[1:] or ['0', '2', '3', '4', '5']:
    print('TELL_GEMM_TILE=%s' % v, flush=True)
    subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, TELL_GEMM_TILE=v))
