"""Does de-synchronising the epilogues of the CUs help?  PROBE BUILD REQUIRED - not runnable against the committed
kernel: in gemm_nt_pp_kernel let odd workgroups spin (s_sleep) for -p.alpha x ~3.5 us before their main loop when
p.alpha < 0, rebuild, run, revert.  Recorded result: no stagger is as fast as any (qkv 131 us flat) - the epilogue is
bound by per-CU store issue, not by an HBM write burst."""
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
for name, N, K, act in (('qkv', 3072, 1024, 0), ('fc1+gelu', 4096, 1024, 2), ('out', 1024, 1024, 0)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda'); y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    res = [timed(lambda: ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act))]
    for n in (1, 2, 3, 4, 6):
        res.append(timed(lambda: ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act, alpha=-float(n))))
    print('%-9s no stagger %.1f us | stagger 1,2,3,4,6 x ~3.5us: %s' % (name, res[0], ' '.join('%.1f' % r for r in res[1:])))
