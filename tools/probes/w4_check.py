import sys, math, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import ops, hip
ok = True
for M, N, K in [(4096, 4096, 64), (4096, 4096, 128), (4096, 4096, 192), (8192, 2048, 320), (2048, 8192, 1024), (16384, 1024, 4096)]:
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.randn(M, K, generator=g).bfloat16(); b = torch.randn(N, K, generator=g).bfloat16()
    bias_n = torch.randn(N, generator=g)
    ad, bd = a.cuda(), b.cuda()
    ref = (ad.float() @ bd.float().t())
    out = ops.gemm(ad, bd)
    plan = hip.query('tell_gemm_nt_plan', ad, ad.stride(0), bd, bd.stride(0), out, out.stride(0), M, N, K, hip.BF16, hip.BF16, None, 0, 0, None, 1.0, 0, None)
    err = ((out.float() - ref).norm() / ref.norm()).item()
    out2 = ops.gemm(ad, bd, bias=bias_n.cuda(), bias_mode=1, act=2, alpha=0.5)
    ref2 = torch.nn.functional.gelu((ref + bias_n.cuda()) * 0.5)
    err2 = ((out2.float() - ref2).norm() / ref2.norm()).item()
    same = all(torch.equal(ops.gemm(ad, bd), out) for _ in range(4))
    print(plan, M, N, K, 'rel err %.2e %.2e' % (err, err2), 'repeatable', same)
    ok = ok and err < 4e-3 and err2 < 4e-3 and same
print('OK' if ok else 'FAILED')
