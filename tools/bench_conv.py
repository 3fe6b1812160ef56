"""Device time of the implicit-GEMM convolution (tell_conv_bn_stats without the finish) on the ResNet-152 bottleneck shapes at
B = 32, each captured as 10 launches in one hipGraph, per tile shape and LDS ring depth (TELL_CONV_TILE x TELL_GEMM_RING), and
of the default choice with and without the BatchNorm launch behind it."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [('l1 conv1', 56, 256, 1, 1, 64), ('l1 conv2', 56, 64, 3, 1, 64), ('l1 conv3', 56, 64, 1, 1, 256),
          ('l2 conv1', 28, 512, 1, 1, 128), ('l2 conv2', 28, 128, 3, 1, 128), ('l2 conv3', 28, 128, 1, 1, 512),
          ('l3 conv1', 14, 1024, 1, 1, 256), ('l3 conv2', 14, 256, 3, 1, 256), ('l3 conv3', 14, 256, 1, 1, 1024),
          ('l4 conv1', 7, 2048, 1, 1, 512), ('l4 conv2', 7, 512, 3, 1, 512), ('l4 conv3', 7, 512, 1, 1, 2048),
          ('l3 ds', 28, 512, 1, 2, 1024), ('l3 conv2 s2', 28, 256, 3, 2, 256)]
COMBOS = [('1', '2'), ('2', '2'), ('3', '2'), ('3', '4')]      # (tile, LDS stages) - the instantiated ones; round 5 also measured
# 128x128 x 4, 128x64 x 4 / 6, 64x64 x 8 stages (profiles/r05_conv_shapes.txt): none won, removed from the library
print('%-12s %7s %5s %5s | us conv + statistics epilogue (no finish) per (tile, stages): 1 = 128x128, 2 = 128x64, 3 = 64x64 | default choice' % ('shape', 'M', 'N', 'K'))
print('%-12s %7s %5s %5s | %s | default' % ('', '', '', '', ' '.join('%7s' % ('%s/%s' % c) for c in COMBOS)))
for name, H, Cin, k, s, Cout in shapes:
    p = k // 2
    OH = (H + 2 * p - k) // s + 1
    M, K = B * OH * OH, k * k * Cin
    x = torch.randn(B, H, H, Cin, device='cuda').bfloat16()
    w = (torch.randn(Cout, K, device='cuda') * 0.05).bfloat16()
    y = torch.empty(M, Cout, dtype=torch.bfloat16, device='cuda')
    gamma = torch.ones(Cout, device='cuda'); beta = torch.zeros(Cout, device='cuda')
    rm = torch.zeros(Cout, device='cuda'); rv = torch.ones(Cout, device='cuda')
    out = []

    def conv_only():      # statistics epilogue on (workspace given), the finish launch not issued: mean = None keeps it a plain launch
        hip.call('tell_conv_bn_stats', x, w, y, B, H, H, Cin, k, k, s, p, OH, OH, Cout, 1e-5, 0.1, None, None, None, None, None, zero)
    for tile, ring in COMBOS:
        hip.apply_env({'TELL_CONV_TILE': tile, 'TELL_GEMM_RING': ring})
        out.append('%7.1f' % timed(conv_only))
    hip.apply_env({'TELL_CONV_TILE': None, 'TELL_GEMM_RING': None})
    conv_only(); torch.cuda.synchronize(); y_ref = y.clone()
    for mode in ('0', '2'):                   # gemm_s64.hip: 0 = never (the general body), 2 = always (default: from K = 1024)
        hip.apply_env({'TELL_GEMM_S64': mode})
        t_s = timed(conv_only)
        y.zero_(); conv_only(); torch.cuda.synchronize()
        same = bool((y.float() - y_ref.float()).abs().max() <= 0.02 * y_ref.float().abs().max())
        tf_s = timed(lambda: hip.call('tell_conv_bn_act', x, w, y, B, H, H, Cin, k, k, s, p, OH, OH, Cout, 1e-5, 0.1, gamma, beta, rm,
                                      rv, None, 1, ws, zero))
        out.append('| s64/%s %6.1f (%s) +BN %6.1f' % (mode, t_s, 'ok' if same else 'DIFF', tf_s))
    hip.apply_env({'TELL_GEMM_S64': None})
    td = timed(conv_only)
    tf = timed(lambda: hip.call('tell_conv_bn_act', x, w, y, B, H, H, Cin, k, k, s, p, OH, OH, Cout, 1e-5, 0.1, gamma, beta, rm, rv,
                                None, 1, ws, zero))
    print('%-12s %7d %5d %5d | %s | %6.1f us = %4.0f TFLOP/s; conv + BN + ReLU %6.1f us' % (
        name, M, Cout, K, ' '.join(out), td, 2.0 * M * Cout * K * 1e-6 / td, tf), flush=True)
