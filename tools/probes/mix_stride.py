"""weigh_bert mix over the 25-layer RoBERTa stack (transformer_faces_objects.py:355-364): the layers are n = B*512*1024 bf16 =
exactly 32 MB apart, so the 25 reads of one output position sit on one HBM channel.  Times tell_mix_fwd / tell_mix_bwd at
n and at n + pad (the pad is simply mixed along: same kernel, layer pitch no longer a power of two)."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip
hip.require_gpu()
L, n0 = 25, 32 * 512 * 1024
junk = torch.empty(160 * 1024 * 1024, device='cuda')


def timed(fn, reps=6):
    def body():
        hip.call('tell_fill_f32', junk, junk.numel(), 1.0)      # cold caches, as in the step
        fn()
    def flush():
        hip.call('tell_fill_f32', junk, junk.numel(), 1.0)
    res = []
    for f in (body, flush):
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), hip.bound_stream():
            for _ in range(reps):
                f()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            if r >= 1:
                ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        res.append(statistics.median(ts))
    return res[0] - res[1]


for pad in (0, 4096, 65536, 1024 * 1024 + 4096):
    n = n0 + pad
    H = torch.randn(L, n, device='cuda').bfloat16(); w = torch.randn(L, device='cuda')
    out = torch.empty(n, device='cuda', dtype=torch.bfloat16); dout = torch.randn(n, device='cuda').bfloat16()
    nb = 2048
    partial = torch.empty(nb * L, device='cuda')
    tf = timed(lambda: hip.call('tell_mix_fwd', H, w, L, n, out, hip.BF16))
    tb = timed(lambda: hip.call('tell_mix_bwd', H, dout, L, n, partial, nb, hip.BF16))
    by = L * n * 2
    print('layer pitch n0 + %7d elements: fwd %6.1f us (%.2f TB/s)   bwd %6.1f us (%.2f TB/s)' % (pad, tf, (by + 2 * n) / tf * 1e-6, tb, (by + 2 * n) / tb * 1e-6))
