"""fc1 (bias + erf-GELU epilogue) and the other RoBERTa GEMM shapes under each tile dispatch (TELL_GEMM_TILE),
interleaved rounds inside one process per setting.  Measurement aid."""
import os, subprocess, sys
CHILD = r'''
import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import ops
shapes = [(8192, 4096, 1024, 2), (8192, 4096, 1024, 0), (8192, 3072, 1024, 0), (8192, 1024, 4096, 0), (8192, 1024, 1024, 0), (8192, 2048, 1024, 0)]
for M, N, K, act in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda')
    out = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
    f = lambda: ops.gemm(a, b, out=out, bias=bias, bias_mode=1, act=act)
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = []
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20; e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / n)
    us = sorted(best)[1]
    print('  M%-5d N%-5d K%-5d act%d %8.1f us %7.1f TF/s' % (M, N, K, act, us, 2.0 * M * N * K / us / 1e6), flush=True)
'''
for v in sys.argv[1:] or ['0', '1', '8']:
    print('TELL_GEMM_TILE=%s' % v, flush=True)
    subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, TELL_GEMM_TILE=v))
