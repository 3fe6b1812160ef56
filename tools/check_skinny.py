#!/usr/bin/env python
"""Numerical check of csrc/decode.hip entry points against torch (GPU box): python tools/check_skinny.py [M]"""
import sys

import torch
import torch.nn.functional as Fn

sys.path.insert(0, '.')
import tell_amd  # noqa: E402
from tell_amd import decode, ops  # noqa: E402

tell_amd.hip.require_gpu()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
E, F = 1024, 4096
dev = 'cuda'
bf = dict(dtype=torch.bfloat16, device=dev)
f32 = dict(dtype=torch.float32, device=dev)
torch.manual_seed(0)


class LN:
    def __init__(self):
        self.weight = torch.rand(E, **f32) + 0.5
        self.bias = torch.randn(E, **f32) * 0.1
        self.eps = 1e-5


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


W = lambda n, k: torch.randn(n, k, **bf) * 0.03      # noqa: E731
B = lambda n: torch.randn(n, **f32) * 0.1            # noqa: E731
x = torch.randn(M, E, **bf)
x4 = torch.randn(M, F, **bf)
raw = torch.randn(M, E, **f32) * 2 + 0.3
raw4 = torch.randn(M, 4 * E, **f32) * 2 - 0.2
ln, lns = LN(), [LN() for _ in range(4)]
lnf = lambda t, l: Fn.layer_norm(t, (E,), l.weight, l.bias, l.eps)     # noqa: E731

# GLU pro0 / pro1
w, b = W(2 * E, E), B(2 * E)
g = torch.empty(M, E, **bf)
decode._skinny([x], E, [w], [b], [g], E, M, E, E, act=2)
print('linear1+GLU pro0   %.2e' % rel(g, Fn.glu(x.float() @ w.float().t() + b, dim=-1)))
st = torch.zeros(M, 2, **f32)
decode._skinny([raw], E, [w], [b], [g], E, M, E, E, pro=1, gammas=[ln.weight], betas=[ln.bias], stats_out=st, act=2)
xn = lnf(raw, ln)
print('linear1+GLU pro1   %.2e' % rel(g, Fn.glu(xn.bfloat16().float() @ w.float().t() + b, dim=-1)))
print('   stats mean %.2e rstd %.2e' % (rel(st[:, 0], raw.mean(1)), rel(st[:, 1], (raw.var(1, unbiased=False) + 1e-5).rsqrt())))
# linear2 + res (bf16), + res_raw
w, b = W(E, E), B(E)
o = torch.empty(M, E, **f32)
decode._skinny([x], E, [w], [b], [o], E, M, E, E, res=x, ld_res=E, out_f32=True)
print('linear2+res        %.2e' % rel(o, x.float() @ w.float().t() + b + x.float()))
decode._skinny([x], E, [w], [b], [o], E, M, E, E, res_raw=raw, res_stats=st, res_ln=ln, out_f32=True)
print('linear2+LN(res)    %.2e' % rel(o, x.float() @ w.float().t() + b + xn))
# q-proj x4
wq, bq = [W(E, E) for _ in range(4)], [B(E) for _ in range(4)]
q4 = torch.empty(4, M, E, **bf)
decode._skinny([raw] * 4, E, wq, bq, [q4[i] for i in range(4)], E, M, E, E, pro=1, gammas=[ln.weight], betas=[ln.bias], stats_out=st, scale=0.125)
print('q-proj x4          ' + ' '.join('%.2e' % rel(q4[i], (xn.bfloat16().float() @ wq[i].float().t() + bq[i]) * 0.125) for i in range(4)))
# out-proj x4
r6 = torch.empty(M, 4 * E, **f32)
decode._skinny([q4[i] for i in range(4)], E, wq, bq, [r6[:, i * E:(i + 1) * E] for i in range(4)], 4 * E, M, E, E, res_raw=raw, res_stats=st, res_ln=ln, out_f32=True)
print('out-proj x4        ' + ' '.join('%.2e' % rel(r6[:, i * E:(i + 1) * E], q4[i].float() @ wq[i].float().t() + bq[i] + xn) for i in range(4)))
# context_fc
wc, bc = W(E, 4 * E), B(E)
decode._skinny([raw4], 4 * E, [wc], [bc], [g], E, M, E, 4 * E, pro=2, gammas=[l.weight for l in lns], betas=[l.bias for l in lns], seg=E)
cat = torch.cat([lnf(raw4[:, i * E:(i + 1) * E], lns[i]) for i in range(4)], 1)
print('context_fc pro2    %.2e' % rel(g, cat.bfloat16().float() @ wc.float().t() + bc))
# fc1 / fc2
w1, b1, w2, b2 = W(F, E), B(F), W(E, F), B(E)
h = torch.empty(M, F, **bf)
decode._skinny([x], E, [w1], [b1], [h], F, M, F, E, act=1)
print('fc1+relu           %.2e' % rel(h, torch.relu(x.float() @ w1.float().t() + b1)))
decode._skinny([x4], F, [w2], [b2], [o], E, M, E, F, res=x, ld_res=E, out_f32=True)
print('fc2+res            %.2e' % rel(o, x4.float() @ w2.float().t() + b2 + x.float()))
# dynconv step
for K in (3, 31):
    H = 16
    wt = W(H * K, E)
    hist = torch.randn(K, M, E, **bf)                # ring of K planes; step t = K - 1 reads planes K-2 .. 0, writes plane K-1
    h0 = hist.clone()
    y = torch.empty(M, E, **bf)
    ops.call('tell_dynconv_step', x, hist, wt, y, M, E, H, K, K - 1, None)
    taps = torch.softmax((x.float() @ wt.float().t()).view(M, H, K), -1)
    win = torch.cat([h0[:K - 1], x[None]], 0).float().view(K, M, H, 64)
    want = torch.einsum('mhk,kmhd->mhd', taps, win).reshape(M, E)
    print('dynconv K=%-2d       %.2e   ring plane ok: %s' % (K, rel(y, want), torch.equal(hist, torch.cat([h0[:K - 1], x[None]], 0))))
