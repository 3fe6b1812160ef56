"""RoBERTa sub-layer tail  LayerNorm(residual + dropout(linear(x)))  at B x 512 = 16384 rows, bf16, p = 0.1:
GEMM + LayerNorm(dropout(x) + residual) against GEMM-with-residual-epilogue (tell_gemm_nt_dropout_residual) + LayerNorm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


M, E, p = 16384, 1024, 0.1
gam, bet = torch.ones(E, device='cuda'), torch.zeros(E, device='cuda')
for name, K in (('out_proj', 1024), ('fc2', 4096)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(E, K, device='cuda').bfloat16()
    bias = torch.randn(E, device='cuda'); res = torch.randn(M, E, device='cuda').bfloat16()
    y = torch.empty(M, E, device='cuda', dtype=torch.bfloat16); o = torch.empty_like(y)

    def unfused():
        ops.gemm(a, w, out=y, bias=bias, bias_mode=1)
        hip.call('tell_layernorm_fwd', y, E, res, E, gam, bet, o, E, None, None, M, E, 1e-5, p, 1, 2, hip.BF16)

    def fused():
        assert hip.call_rc('tell_gemm_nt_dropout_residual', a, K, w, K, bias, res, E, y, E, M, E, K, p, 1, 2) == 0
        hip.call('tell_layernorm_fwd', y, E, None, 0, gam, bet, o, E, None, None, M, E, 1e-5, 0.0, 0, 0, hip.BF16)

    def gemm_only():
        ops.gemm(a, w, out=y, bias=bias, bias_mode=1)

    def fused_gemm_only():
        hip.call_rc('tell_gemm_nt_dropout_residual', a, K, w, K, bias, res, E, y, E, M, E, K, p, 1, 2)
    print('%-8s K=%4d  gemm %6.1f + LN(res, dropout) = %6.1f us | gemm(res, dropout) %6.1f + LN = %6.1f us'
          % (name, K, timed(gemm_only), timed(unfused), timed(fused_gemm_only), timed(fused)))
