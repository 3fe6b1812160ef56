"""Weight-gradient products dW = dY^T X of the decoder (K-major operands read in place, gemm_tx kernels): operand row
pitches as the forward pass leaves them (powers of two: 2 / 4 / 8 KB) against the same with 64 / 256 extra elements per
row, caches flushed between launches as inside the step."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()
junk = torch.empty(160 * 1024 * 1024, device='cuda')


def timed(fn, reps=6, flush=True):
    def fl():
        hip.call('tell_fill_f32', junk, junk.numel(), 1.0)
    res = []
    for body in ((lambda: (fl() if flush else None, fn())), (lambda: fl() if flush else None)):
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), hip.bound_stream():
            for _ in range(reps):
                body()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            if r >= 1:
                ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        res.append(statistics.median(ts))
    return res[0] - res[1]


R = 1024
for n_out, n_in in ((4096, 1024), (1024, 4096), (2048, 1024), (1024, 1024)):
    line = 'dW [%d x %d] from %d rows:' % (n_out, n_in, R)
    for pad in (0, 64, 256):
        dy = torch.randn(R, n_out + pad, device='cuda').bfloat16()[:, :n_out]
        x = torch.randn(R, n_in + pad, device='cuda').bfloat16()[:, :n_in]
        out = torch.zeros(n_out, n_in, device='cuda')
        # four such products per launch (one per decoder layer), as the trainer's queue sends them
        outs = [torch.zeros(n_out, n_in, device='cuda') for _ in range(4)]
        def fn():
            ops.gemm_grouped([dict(a=dy, b=x, out=o, form='tn', accumulate=False) for o in outs])
        t = timed(fn)
        line += '  pitch +%3d: %6.1f us (%4.0f TFLOP/s)' % (pad, t, 4 * 2.0 * R * n_out * n_in / t * 1e-6)
    print(line)
