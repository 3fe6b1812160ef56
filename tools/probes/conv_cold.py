"""How much of a trunk convolution's time is cold operands?  layer3 conv2 / conv1 / conv3 at B = 32 (default kernels):
back to back (operands in L2), behind a 64 MB fill (L2 flushed, MALL warm), behind a 1 GB fill (MALL flushed too), and
behind a fill + a PREFETCH kernel that only reads the weights (every XCD pulls them into its L2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip
zero = torch.zeros(256, dtype=torch.uint8, device='cuda')
small = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
big = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
B = 32
for name, H, Cin, k, s, Cout in (('l3 conv2', 14, 256, 3, 1, 256), ('l3 conv1', 14, 1024, 1, 1, 256), ('l3 conv3', 14, 256, 1, 1, 1024),
                                 ('l4 conv2', 7, 512, 3, 1, 512)):
    p = k // 2
    OH = (H + 2 * p - k) // s + 1
    M, K = B * OH * OH, k * k * Cin
    x = torch.randn(B, H, H, Cin, device='cuda').bfloat16()
    w = (torch.randn(Cout, K, device='cuda') * 0.05).bfloat16()
    y = torch.empty(M, Cout, dtype=torch.bfloat16, device='cuda')

    def conv():
        hip.call('tell_conv_bn_stats', x, w, y, B, H, H, Cin, k, k, s, p, OH, OH, Cout, 1e-5, 0.1, None, None, None, None, None, zero)

    def timed(pre):
        ts = []
        for _ in range(12):
            pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); conv(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    t_hot = timed(lambda: None)
    t_l2 = timed(lambda: small.fill_(1))
    t_mall = timed(lambda: big.fill_(1))
    t_pf_w = timed(lambda: (small.fill_(1), w.sum()))          # a reduction over w: SOME XCDs' L2s (and the MALL) hold it again
    t_pf_x = timed(lambda: (small.fill_(1), x.sum()))
    print('%-9s M %5d N %4d K %4d (x %.1f MB, w %.1f MB): hot %5.1f us | L2 flushed %5.1f | MALL flushed %5.1f | L2 flushed then w re-read %5.1f, x re-read %5.1f'
          % (name, M, Cout, K, x.numel() * 2e-6, w.numel() * 2e-6, t_hot, t_l2, t_mall, t_pf_w, t_pf_x), flush=True)
