"""gemm_nt_q4_kernel (TELL_GEMM_Q4=1) against an fp32 product of the same bf16 operands: K tile counts 2, 4, 6, 16, 64 walk
first / steady-state / last bodies of the hand-placed loop and the cross-tile DMA stream (1, 2, 3 and 4 output tiles per
resident workgroup, also a last round with fewer tiles than workgroups), every epilogue form, strided operands, repeated launches
bit-identical.  Run with TELL_GEMM_Q4=1; prints the kernel the library planned for each shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()
ok = True
SHAPES = ((16384, 1024, 128), (16384, 2048, 256), (16384, 3072, 384), (16384, 4096, 1024), (16384, 1024, 4096),
          (8192, 8192, 128), (4096, 4096, 256), (16384, 2048, 640), (16384 + 256 * 8, 1024, 256), (16384, 3072, 1024), (16384, 2048, 2048),
          # variable-length batches (B x L = 12288 / 8192 rows): partial rounds at least 70 % full stay on q4 by default
          (12288, 1024, 1024), (12288, 1024, 4096), (12288, 2048, 1024), (8192, 3072, 1024), (12288, 3072, 1024))
for M, N, K in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    b = torch.randn(N, K, device='cuda', generator=g).bfloat16()
    bn = torch.randn(N, device='cuda', generator=g); bm = torch.randn(M, device='cuda', generator=g)
    ref = a.float() @ b.float().t()
    y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    plan = hip.query('tell_gemm_nt_plan', a, a.stride(0), b, b.stride(0), y, y.stride(0), M, N, K, hip.BF16, hip.BF16, None, 0, 0, None, 1.0, 0, None)
    o1 = ops.gemm(a, b)
    o2 = ops.gemm(a, b, bias=bn, bias_mode=1, act=2, alpha=0.5)
    o3 = ops.gemm(a, b, bias=bm, bias_mode=2, act=1)
    # per-column bias: gemm_nt_q4e_kernel where it applies (K >= 576, whole rounds) - every activation
    plan_e = hip.query('tell_gemm_nt_plan', a, a.stride(0), b, b.stride(0), y, y.stride(0), M, N, K, hip.BF16, hip.BF16, bn, 1, 0, None, 1.0, 0, None)
    o4 = ops.gemm(a, b, bias=bn, bias_mode=1)
    o5 = ops.gemm(a, b, bias=bn, bias_mode=1, act=1, alpha=2.0)
    e4 = ((o4.float() - (ref + bn)).norm() / (ref + bn).norm()).item()
    r5 = torch.relu((ref + bn) * 2.0); e5 = ((o5.float() - r5).norm() / r5.norm()).item()
    same_e = all(torch.equal(ops.gemm(a, b, bias=bn, bias_mode=1), o4) and
                 torch.equal(ops.gemm(a, b, bias=bn, bias_mode=1, act=2, alpha=0.5), o2) for _ in range(3))
    e1 = ((o1.float() - ref).norm() / ref.norm()).item()
    r2 = torch.nn.functional.gelu((ref + bn) * 0.5); e2 = ((o2.float() - r2).norm() / r2.norm()).item()
    r3 = torch.relu(ref + bm[:, None]); e3 = ((o3.float() - r3).norm() / r3.norm()).item()
    worst = (o1.float() - ref).abs().max().item() / ref.abs().max().item()
    same = all(torch.equal(ops.gemm(a, b), o1) for _ in range(3))
    good = max(e1, e2, e3, e4, e5) < 4e-3 and same and same_e and worst < 2e-2
    ok &= good
    print('%-32s %-32s M=%5d N=%4d K=%4d  err %.2e %.2e %.2e %.2e %.2e  max %.2e  repeat-identical %s %s  %s'
          % (plan, plan_e, M, N, K, e1, e2, e3, e4, e5, worst, same, same_e, 'ok' if good else 'FAIL'))
# strided operands (a column slice of a wider matrix; output into a column slice): the packed QKV / K|V layouts
M, N, K = 16384, 1024, 1024
g = torch.Generator(device='cuda').manual_seed(5)
big_a = torch.randn(M, 2 * K, device='cuda', generator=g).bfloat16(); a = big_a[:, K:]
big_b = torch.randn(N, 3 * K, device='cuda', generator=g).bfloat16(); b = big_b[:, K:2 * K]
big_y = torch.zeros(M, 2 * N, device='cuda', dtype=torch.bfloat16); y = big_y[:, N:]
bs = torch.randn(N, device='cuda', generator=g)
ops.gemm(a, b, out=y, bias=bs, bias_mode=1)
ref = a.float() @ b.float().t() + bs
e = ((y.float() - ref).norm() / ref.norm()).item()
untouched = bool((big_y[:, :N] == 0).all())
print('strided operands / output: err %.2e, neighbour columns untouched %s' % (e, untouched))
ok &= e < 4e-3 and untouched
print('ALL OK' if ok else 'FAILED')
