import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import ops
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
# many independent launches back to back: measures device time per launch when the host keeps up
for M, N, K in [(512, 1024, 1024), (1024, 1024, 512), (512, 4096, 1024), (512, 1024, 4096), (3136, 256, 1024), (3136, 1024, 256), (784, 512, 4608)]:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    akm = a.t().contiguous(); bkn = b.t().contiguous()
    o16 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); o32 = torch.zeros(M, N, device='cuda')
    print('M%-5d N%-5d K%-5d  NT %6.1f us | NN %6.1f | TN bf16 %6.1f | TN f32 acc %6.1f | NT f32 acc %6.1f' % (
        M, N, K, t(lambda: ops.gemm(a, b, out=o16)), t(lambda: ops.gemm_nn(a, bkn, out=o16)),
        t(lambda: ops.gemm_tn(akm, bkn, out=o16)), t(lambda: ops.gemm_tn(akm, bkn, out=o32, accumulate=True)),
        t(lambda: ops.gemm(a, b, out=o32, accumulate=True))))
