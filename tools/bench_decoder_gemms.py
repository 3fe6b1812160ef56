"""Device time of the decoder's GEMM shapes at BASELINE configs[2] (T*B = 1024 rows, E = 1024), each form captured as
20 launches in one hipGraph (no host time in the numbers): forward NT, dgrad (NN), wgrad (TN, fp32 accumulate)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import hip, ops
tell_amd.set_compute_dtype(torch.bfloat16)
REP = 20


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * REP)


rows = [('linear1 / kv-faces', 1024, 2048, 1024), ('q/out/linear2', 1024, 1024, 1024), ('fc1', 1024, 4096, 1024),
        ('fc2 / context_fc', 1024, 1024, 4096), ('taps K=31', 1024, 496, 1024), ('kv article', 16384, 2048, 1024),
        ('kv image', 1568, 2048, 2048), ('kv obj', 2048, 2048, 2048), ('kv faces', 128, 2048, 512),
        ('head', 1024, 5002, 1024), ('tail2 logits', 1024, 30265, 1024)]
print('%-20s %6s %6s %6s | %9s %9s %9s   (us; TFLOP/s)' % ('shape', 'M', 'N', 'K', 'fwd NT', 'dgrad NN', 'wgrad TN'))
for name, M, N, K in rows:
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    dy = torch.randn(M, N, device='cuda').bfloat16()
    y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); dx = torch.empty(M, K, device='cuda', dtype=torch.bfloat16)
    gw = torch.zeros(N, K, device='cuda')
    Kp = (N + 7) // 8 * 8
    if Kp != N:
        dyp = torch.zeros(M, Kp, device='cuda', dtype=torch.bfloat16); dyp[:, :N] = dy; dy = dyp[:, :N]
    f = 2.0 * M * N * K * 1e-6
    t1 = timed(lambda: ops.gemm(a, w, out=y))
    t2 = timed(lambda: ops.gemm_nn(dyp if Kp != N else dy, w, out=dx))
    t3 = timed(lambda: ops.gemm_tn(dy, a, out=gw, accumulate=True))
    print('%-20s %6d %6d %6d | %5.1f %4.0f %5.1f %4.0f %5.1f %4.0f' % (name, M, N, K, t1, f / t1, t2, f / t2, t3, f / t3))
