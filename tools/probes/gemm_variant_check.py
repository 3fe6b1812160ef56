"""GEMM kernel variants behind environment switches, against an fp32 product of the same bf16 operands:
  (default)            gemm_nt_pp2_kernel  - resident 256x256 ping-pong workgroups, next tile prefetched under the epilogue
  TELL_GEMM_PP2=0      gemm_nt_pp_kernel   - one workgroup per tile
K-tile counts 1, 2, 3, 5 (, 16, 32) walk prologue / steady state / tail of the counted-vmcnt pipelines, several tiles per
resident workgroup; every epilogue form; repeated launches bit-identical.  tests/test_gpu_ops.py runs it per switch."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()        # (registers the tile-counter buffer of the resident GEMM launches)
ok = True
SHAPES = ((16384, 4096, 64), (16384, 4096, 128), (16384, 4096, 192), (16384, 4096, 320), (8192, 8192, 256), (16384, 3072, 1024), (16384, 1024, 1024))
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
    e1 = ((o1.float() - ref).norm() / ref.norm()).item()
    r2 = torch.nn.functional.gelu((ref + bn) * 0.5); e2 = ((o2.float() - r2).norm() / r2.norm()).item()
    r3 = torch.relu(ref + bm[:, None]); e3 = ((o3.float() - r3).norm() / r3.norm()).item()
    same = all(torch.equal(ops.gemm(a, b), o1) for _ in range(3))
    good = max(e1, e2, e3) < 4e-3 and same
    ok &= good
    print('%-32s M=%5d N=%4d K=%4d  err %.2e %.2e %.2e  repeat-identical %s  %s' % (plan, M, N, K, e1, e2, e3, same, 'ok' if good else 'FAIL'))
print('ALL OK' if ok else 'FAILED')
