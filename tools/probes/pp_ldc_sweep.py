"""Does the cost of the pp GEMM's output stores depend on the row stride of C?  (qkv: N=3072 pays 10 us per wave of
tiles for its stores, fc1 at N=4096 pays 4.)"""
import sys, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import hip
def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 50
M, K = 16384, 1024
a = torch.randn(M, K, device='cuda').bfloat16()
for N in (1024, 2048, 3072, 4096, 5120):
    w = torch.randn(N, K, device='cuda').bfloat16(); bias = torch.randn(N, device='cuda')
    row = []
    for ldc in (N, N + 64, N + 128, N + 256, 4096 if N < 4096 else 8192):
        y = torch.empty(M, ldc, device='cuda', dtype=torch.bfloat16)
        t = timed(lambda: hip.call('tell_gemm_nt', a, a.stride(0), w, w.stride(0), y, ldc, M, N, K, 1, 1, bias, 1, 0, None, 1.0, 0, None))
        row.append('ldc %5d: %6.1f us (%4.0f TF)' % (ldc, t, 2.0 * M * N * K / t * 1e-6))
    print('N %4d | ' % N + ' | '.join(row))
