"""Is bn_finish_apply_kernel (64-channel slabs: 128-byte pieces one row pitch apart) starved by a power-of-two row pitch?
conv (1x1, Cin 256, 14x14, B = 32) + BatchNorm + ReLU minus the convolution alone, for Cout = 1024 (pitch 2048 B) and
neighbours that are not powers of two."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip
REP = 10
ws = torch.empty(1 << 24, dtype=torch.float32, device='cuda')
zero = torch.zeros(256, dtype=torch.uint8, device='cuda')


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


B, H, Cin = 32, 14, 256
for Cout in (1024, 1088, 960, 2048, 2112, 512, 576, 256, 320):
    M = B * H * H
    x = torch.randn(B, H, H, Cin, device='cuda').bfloat16()
    w = (torch.randn(Cout, Cin, device='cuda') * 0.05).bfloat16()
    y = torch.empty(M, Cout, dtype=torch.bfloat16, device='cuda')
    res = torch.randn(M, Cout, device='cuda').bfloat16()
    gamma = torch.ones(Cout, device='cuda'); beta = torch.zeros(Cout, device='cuda')
    rm = torch.zeros(Cout, device='cuda'); rv = torch.ones(Cout, device='cuda')
    t0 = timed(lambda: hip.call('tell_conv_bn_stats', x, w, y, B, H, H, Cin, 1, 1, 1, 0, H, H, Cout, 1e-5, 0.1, None, None, None, None, ws, zero))
    t1 = timed(lambda: hip.call('tell_conv_bn_act', x, w, y, B, H, H, Cin, 1, 1, 1, 0, H, H, Cout, 1e-5, 0.1, gamma, beta, rm, rv, None, 1, ws, zero))
    t2 = timed(lambda: hip.call('tell_conv_bn_act', x, w, y, B, H, H, Cin, 1, 1, 1, 0, H, H, Cout, 1e-5, 0.1, gamma, beta, rm, rv, res, 1, ws, zero))
    mb = M * Cout * 2e-6
    print('Cout %5d (pitch %5d B, tensor %5.1f MB): conv %5.1f us, + BN + ReLU %5.1f (BN %5.1f us = %.2f TB/s), + residual %5.1f (BN %5.1f us = %.2f TB/s)'
          % (Cout, Cout * 2, mb, t0, t1, t1 - t0, 2 * mb / (t1 - t0) / 1e6 * 1e6 / 1e6, t2, t2 - t0, 3 * mb / (t2 - t0)), flush=True)
