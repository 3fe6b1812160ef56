"""The dominant GEMM kernel's launches of one configs[2] training step, three times each, and nothing else - the command
bench.py runs under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to put a MEASURED HBM traffic
figure into its roofline block: gemm_nt_q4_kernel<bf16,256,256> runs RoBERTa's out-proj / fc1 + GELU / fc2 (24 each per
step) and the side-by-side article K|V projection of the four decoder layers (1 per step).
Prints one line per launch group: name M N K count (dispatch order = this order)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()
tell_amd.set_compute_dtype(torch.bfloat16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 512
MIX = [('out', 1024, 1024, 0, True, 24), ('fc1+gelu', 4096, 1024, 2, True, 24), ('fc2', 1024, 4096, 0, True, 24),
       ('kv_article_x4', 8192, 1024, 0, False, 1)]
g = torch.Generator(device='cuda').manual_seed(0)
for name, N, K, act, has_bias, per_step in MIX:
    a = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    w = (torch.randn(N, K, device='cuda', generator=g) * 0.03).bfloat16()
    bias = torch.randn(N, device='cuda', generator=g) if has_bias else None
    y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    plan = hip.query('tell_gemm_nt_plan', a, a.stride(0), w, w.stride(0), y, y.stride(0), M, N, K, hip.BF16, hip.BF16, bias,
                     1 if has_bias else 0, act, None, 1.0, 0, None)
    for _ in range(3):
        ops.gemm(a, w, out=y, bias=bias, bias_mode=1 if has_bias else 0, act=act)
    torch.cuda.synchronize()
    print('MIX %s %d %d %d %d %s' % (name, M, N, K, per_step, plan), flush=True)
