import sys, ctypes, torch
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
M = 16384
for name, N, K, act in (('qkv', 3072, 1024, 0), ('out', 1024, 1024, 0), ('fc1+gelu', 4096, 1024, 2)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda'); y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    def run(aux, alpha=1.0):
        hip.call('tell_gemm_nt', a, a.stride(0), w, w.stride(0), y, y.stride(0), M, N, K, 1, 1, bias, 1, act, aux, alpha, 0, None)
    t_full = timed(lambda: run(None))
    t_nostore = timed(lambda: run(1))
    print('%-9s full %.1f us | LDS staging but no global stores %.1f us' % (name, t_full, t_nostore))
