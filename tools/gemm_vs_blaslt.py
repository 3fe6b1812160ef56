"""Headroom check for the dominant kernel: the hand-written NT GEMM next to torch.matmul (hipBLASLt) on the
RoBERTa / context-projection shapes of configs[1].  Measurement only - the product path never calls the library."""
import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import ops

shapes = [(16384, 3072, 1024), (16384, 4096, 1024), (16384, 1024, 4096), (16384, 1024, 1024), (16384, 2048, 1024),
          (8192, 3072, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (8192, 1024, 1024), (8192, 2048, 1024),
          (784, 2048, 2048), (512, 4096, 1024), (512, 1024, 4096), (512, 2048, 1024)]


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16()
    b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
    bt = b.t()
    ours = bench(lambda: ops.gemm(a, b, out=out))
    lib = bench(lambda: torch.matmul(a, bt, out=out))
    fl = 2.0 * M * N * K / 1e6
    print('M%-5d N%-5d K%-5d  ours %7.1f us %7.1f TF/s | hipBLASLt %7.1f us %7.1f TF/s | ratio %.2f'
          % (M, N, K, ours, fl / ours, lib, fl / lib, lib / ours), flush=True)
