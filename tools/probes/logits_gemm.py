"""The three fp32 logits GEMMs of the adaptive softmax at M = 1024 rows, alone (rows padded to 16 bytes: the staged fp32
epilogue of the direct-to-LDS kernel applies).  (64x64 instead of 128x128 tiles for the K <= 128 ones: 27.8 vs 29.8 us - not adopted.)"""
import sys, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import ops, hip
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
for M, N, K in ((1024, 30265, 64), (1024, 15000, 256), (1024, 5002, 1024)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, (N + 3) // 4 * 4, device='cuda')[:, :N]
    t = timed(lambda: ops.gemm(a, w, out=out))
    print('M %d N %5d K %4d fp32 out: %6.1f us  (%.2f TB/s of output)' % (M, N, K, t, M * N * 4 / t * 1e-6))
