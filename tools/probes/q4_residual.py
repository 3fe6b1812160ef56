"""The residual + dropout epilogue (tell_gemm_nt_dropout_residual) on the four-wave kernel vs the ping-pong kernel vs the plain
GEMM of the same shape, interleaved inside one process (RoBERTa out-proj and fc2 at M = 16384)."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()
M, REP, ROUNDS = 16384, 10, 5


def graph_of(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    return g


def time_graph(g):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * REP)


graphs, keep = {}, []          # (a captured graph holds raw pointers: the tensors must outlive it)
for name, N, K in (('out', 1024, 1024), ('fc2', 1024, 4096)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda').bfloat16(); y = torch.empty_like(res)
    for cname, env in (('plain q4', {'TELL_GEMM_Q4': '1'}), ('res q4', {'TELL_GEMM_Q4': '1'}), ('res pp2', {'TELL_GEMM_Q4': '0'})):
        if len(sys.argv) > 1 and cname not in sys.argv[1:]:
            continue
        hip.apply_env(env)
        if cname == 'plain q4':
            fn = lambda a=a, w=w, bias=bias, y=y: ops.gemm(a, w, out=y, bias=bias, bias_mode=1)
        else:
            fn = lambda a=a, w=w, bias=bias, y=y, res=res, N=N, K=K: hip.call('tell_gemm_nt_dropout_residual', a, K, w, K, bias, res, N, y, N, M, N, K, 0.1, 17, 23)
        graphs[(cname, name)] = graph_of(fn)
        keep.append(fn)
times = {k: [] for k in graphs}
for r in range(ROUNDS):
    for k, g in graphs.items():
        times[k].append(time_graph(g))
for k, t in times.items():
    print('%-9s %-4s median %6.1f us (min %6.1f)' % (k[0], k[1], statistics.median(t), min(t)))
