"""How much of a ping-pong GEMM launch is its epilogue?  PROBE BUILD REQUIRED - not runnable against the committed kernel:
in gemm_nt_pp_kernel (csrc/gemm.hip) return before glds_store_tile when p.alpha == -12345.f, rebuild, run, revert.
Recorded result (MI355X, M = 16384): qkv 131 -> 96 us, out 39 -> 29, fc1 165 -> 114, fc2 114 -> 104 without the
epilogue, i.e. 9-13 us per 256x256 tile."""
import sys, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import hip, ops
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
for name, N, K, act in (('qkv', 3072, 1024, 0), ('out', 1024, 1024, 0), ('fc1+gelu', 4096, 1024, 2), ('fc2', 1024, 4096, 0)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda'); y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    t1 = timed(lambda: ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act))
    t0 = timed(lambda: ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act, alpha=-12345.0))
    print('%-9s full %.1f us   without epilogue %.1f us' % (name, t1, t0))
