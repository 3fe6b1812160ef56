"""Decoder article attention (T = 32 queries, S = 512 keys, B = 32, H = 16) forward / backward with COLD caches: between two
launches a 640 MB buffer is rewritten (the 256 MB MALL and the L2s lose K / V / dK / dV), as inside the decoder step where
the projected K | V were written milliseconds earlier.  Reports time per launch = (graph with flush + kernel) - (graph
with the flush alone), next to the warm back-to-back time.  Packed [S, B, 2E] K | V layout as in the step."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip
hip.require_gpu()
B, H, T, S, D, p = 32, 16, 32, 512, 64, 0.1
E = H * D
q = torch.randn(T, B, E, device='cuda').bfloat16()
WIDE = int(os.environ.get('KV_LAYERS', '1'))          # > 1: K | V (and dK | dV) are the column slice of one layer in a buffer that holds
PAD = int(os.environ.get('KV_PAD', '0'))               # extra elements per row (breaks the power-of-two row pitch)
# batch-major rows (b, s) as the encoder leaves the article: [B, S, layers * 2E (+ pad)], seen as [S, B, ...]
kv_all = torch.randn(B, S, WIDE * 2 * E + PAD, device='cuda').bfloat16().transpose(0, 1)
kv = kv_all[..., 2 * E * (WIDE - 1):2 * E * WIDE]; k, v = kv[..., :E], kv[..., E:]
out = torch.empty_like(q); lse = torch.empty(B * H, T, device='cuda')
bk = torch.randn(E, device='cuda').bfloat16(); bv = torch.randn(E, device='cuda').bfloat16()
mask = torch.zeros(B, S, dtype=torch.uint8, device='cuda')
dout = torch.randn_like(q); dq = torch.empty_like(q); dkv_all = torch.empty(B, S, WIDE * 2 * E + PAD, device='cuda', dtype=torch.bfloat16).transpose(0, 1)
dkv = dkv_all[..., 2 * E * (WIDE - 1):2 * E * WIDE]; dk, dv = dkv[..., :E], dkv[..., E:]
dbk = torch.empty(B, E, device='cuda'); dbv = torch.empty(B, E, device='cuda')
junk = torch.empty(160 * 1024 * 1024, device='cuda')          # 640 MB


def fwd():
    hip.call('tell_attn_fwd', q, k, v, out, lse, mask, bk, bv, B, H, T, S, D, q.stride(0), q.stride(1), k.stride(0),
             k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1), 1, p, 1, 2, hip.dt(q))


def bwd():
    hip.call('tell_attn_bwd', q, k, v, out, dout, lse, mask, bk, bv, dq, dk, dv, dbk, dbv, B, H, T, S, D,
             q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0),
             out.stride(1), 1, p, 1, 2, hip.dt(q))


def flush():
    hip.call('tell_fill_f32', junk, junk.numel(), 1.0)


def timed(fns, reps=8):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(reps):
            for f in fns:
                f()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        if r >= 1:
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


fwd()
t_flush = timed([flush])
print('flush alone %.1f us' % t_flush)
print('forward : warm %.1f us   cold %.1f us' % (timed([fwd], 20), timed([flush, fwd]) - t_flush))
print('backward: warm %.1f us   cold %.1f us' % (timed([bwd], 20), timed([flush, bwd]) - t_flush))
