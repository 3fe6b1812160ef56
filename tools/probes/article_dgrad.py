import sys, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import ops, hip
tell_amd.set_compute_dtype(torch.bfloat16)
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
dy = torch.randn(16384, 2048, device='cuda').bfloat16(); w = torch.randn(2048, 1024, device='cuda').bfloat16()
wt = ops.transpose(w)[0]
print('K-major 128x128      %.1f us' % timed(lambda: ops.gemm_nn(dy, w)))
print('NT with cached w^T   %.1f us' % timed(lambda: ops.gemm_nn(dy, w, b_t=lambda: wt)))
print('transpose            %.1f us' % timed(lambda: ops.transpose(w)))
print(hip.query('tell_gemm_nt_plan', dy, dy.stride(0), wt, wt.stride(0), dy, 1024, 16384, 1024, 2048, hip.BF16, hip.BF16, None, 0, 0, None, 1.0, 0, None))
