"""LayerNorm(res + dropout(x)) forward at RoBERTa's shape ([B*512, 1024] bf16) with and without dropout, and the
decoder's ([1024, 1024]): device time per launch (10 per hipGraph) and bytes moved / time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import hip
REP = 10


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * REP)


for rows in (16384, 1024):
    C = 1024
    x = torch.randn(rows, C, device='cuda').bfloat16(); r = torch.randn(rows, C, device='cuda').bfloat16()
    g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda'); y = torch.empty_like(x)
    mean = torch.empty(rows, device='cuda'); rstd = torch.empty(rows, device='cuda')
    for p in (0.1, 0.0):
        for var in ('0', '1', '2'):
            hip.apply_env({'TELL_LN_VAR': var})
            t = timed(lambda: hip.call('tell_layernorm_fwd', x, C, r, C, g, b, y, C, mean, rstd, rows, C, 1e-5, p, 1, 2, hip.BF16))
            print('rows %5d p=%.1f  TELL_LN_VAR=%s  %6.1f us  %5.2f TB/s' % (rows, p, var, t, 3 * rows * C * 2 / t * 1e-6))
    hip.apply_env({'TELL_LN_VAR': None})
