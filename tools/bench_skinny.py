#!/usr/bin/env python
"""Microbenchmark of the generation-step kernels (csrc/decode.hip) at the shapes of one decoder layer.
usage (GPU box): python tools/bench_skinny.py [M]"""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
import tell_amd  # noqa: E402
from tell_amd import hip, decode, ops  # noqa: E402

tell_amd.hip.require_gpu()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
E, F = 1024, 4096
dev = 'cuda'
bf = dict(dtype=torch.bfloat16, device=dev)
f32 = dict(dtype=torch.float32, device=dev)


class LN:
    def __init__(self):
        self.weight = torch.rand(E, **f32) + 0.5
        self.bias = torch.randn(E, **f32) * 0.1
        self.eps = 1e-5


def timeit(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with tell_amd.hip.bound_stream():
            for _ in range(20):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n // 20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (n // 20 * 20)


x = torch.randn(M, E, **bf)
x4 = torch.randn(M, F, **bf)
raw = torch.randn(M, E, **f32)
raw4 = torch.randn(M, 4 * E, **f32)
stats = torch.rand(M, 2, **f32)
ln = LN()
lns = [LN() for _ in range(4)]
W = lambda n, k: torch.randn(n, k, **bf) * 0.03      # noqa: E731
B = lambda n: torch.randn(n, **f32) * 0.1            # noqa: E731
w_l1, b_l1 = W(2 * E, E), B(2 * E)
w_l2, b_l2 = W(E, E), B(E)
wq, bq = [W(E, E) for _ in range(4)], [B(E) for _ in range(4)]
wc, bc = W(E, 4 * E), B(E)
w1, b1 = W(F, E), B(F)
w2, b2 = W(E, F), B(E)
g = torch.empty(M, E, **bf)
o32 = torch.empty(M, E, **f32)
q4 = torch.empty(4, M, E, **bf)
r6 = torch.empty(M, 4 * E, **f32)
h = torch.empty(M, F, **bf)
st = torch.empty(M, 2, **f32)
cases = {
    'linear1+GLU  pro0 N1024 K1024': lambda: decode._skinny([x], E, [w_l1], [b_l1], [g], E, M, E, E, act=2),
    'linear1+GLU  pro1 N1024 K1024': lambda: decode._skinny([raw], E, [w_l1], [b_l1], [g], E, M, E, E, pro=1, gammas=[ln.weight], betas=[ln.bias], stats_out=st, act=2),
    'linear2+res  pro0 N1024 K1024': lambda: decode._skinny([x], E, [w_l2], [b_l2], [o32], E, M, E, E, res=x, ld_res=E, out_f32=True),
    'q-proj x4    pro1 N1024 K1024': lambda: decode._skinny([raw] * 4, E, wq, bq, [q4[i] for i in range(4)], E, M, E, E, pro=1, gammas=[ln.weight], betas=[ln.bias], stats_out=st, scale=0.125),
    'out-proj x4  pro0 N1024 K1024': lambda: decode._skinny([q4[i] for i in range(4)], E, wq, bq, [r6[:, i * E:(i + 1) * E] for i in range(4)], 4 * E, M, E, E, res_raw=raw, res_stats=stats, res_ln=ln, out_f32=True),
    'context_fc   pro2 N1024 K4096': lambda: decode._skinny([raw4], 4 * E, [wc], [bc], [g], E, M, E, 4 * E, pro=2, gammas=[l.weight for l in lns], betas=[l.bias for l in lns], seg=E),
    'fc1+relu     pro0 N4096 K1024': lambda: decode._skinny([x], E, [w1], [b1], [h], F, M, F, E, act=1),
    'fc2+res      pro0 N1024 K4096': lambda: decode._skinny([x4], F, [w2], [b2], [o32], E, M, E, F, res=x, ld_res=E, out_f32=True),
}
raw4_bf = raw4.bfloat16()
wcf, sc_f, cc_f = decode._folded(torch.nn.Parameter(torch.zeros(1, device='cuda')), wc, lns, E)
cases['context_fc   pro4 N1024 K4096 (folded LN)'] = lambda: decode._skinny([raw4_bf], 4 * E, [wcf], [bc], [g], E, M, E, 4 * E, pro=4, gammas=[sc_f], betas=[cc_f], seg=E)
for opt in (0, 2):
    def with_opt(fn, opt=opt):
        def run():
            with hip.options(sk_split=opt):
                fn()
        return run
    cases['fc2+res      (sk_split=%d)' % opt] = with_opt(cases['fc2+res      pro0 N1024 K4096'])
    cases['context_fc   pro4 (sk_split=%d)' % opt] = with_opt(cases['context_fc   pro4 N1024 K4096 (folded LN)'])
back_tab = torch.arange(M, dtype=torch.int32, device='cuda').repeat(30, 1).contiguous()
for K in (3, 7, 15, 31):
    hist = torch.zeros(K, M, E, **bf)
    wt = W(16 * K, E)
    cases['dynconv_step K=%d' % K] = (lambda hist=hist, wt=wt, K=K: ops.call('tell_dynconv_step', x, hist, wt, g, M, E, 16, K, 5, None))
    cases['dynconv_step K=%d + ancestor table' % K] = (lambda hist=hist, wt=wt, K=K: ops.call('tell_dynconv_step', x, hist, wt, g, M, E, 16, K, 5, back_tab))
print('M = %d' % M)
for name, fn in cases.items():
    print('%-34s %7.2f us' % (name, timeit(fn)))
