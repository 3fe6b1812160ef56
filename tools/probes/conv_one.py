"""One bottleneck convolution of the ResNet trunk at B = 32, launched N times (PMC / rocprof target).
usage: conv_one.py <H> <Cin> <k> <stride> <Cout> [n_launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip
H, Cin, k, s, Cout = [int(a) for a in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 20
B = 32
zero = torch.zeros(256, dtype=torch.uint8, device='cuda')
p = k // 2
OH = (H + 2 * p - k) // s + 1
M, K = B * OH * OH, k * k * Cin
x = torch.randn(B, H, H, Cin, device='cuda').bfloat16()
w = (torch.randn(Cout, K, device='cuda') * 0.05).bfloat16()
y = torch.empty(M, Cout, dtype=torch.bfloat16, device='cuda')
other = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
for i in range(n):
    other.fill_(i & 255)                       # (another kernel's traffic in between: the input is not left in any L2)
    hip.call('tell_conv_bn_stats', x, w, y, B, H, H, Cin, k, k, s, p, OH, OH, Cout, 1e-5, 0.1, None, None, None, None, None, zero)
torch.cuda.synchronize()
print('M %d N %d K %d: operands %.1f MB (x %.1f + w %.1f), output %.1f MB, staged through LDS %.1f MB' % (
    M, Cout, K, (x.numel() + w.numel()) * 2e-6, x.numel() * 2e-6, w.numel() * 2e-6, y.numel() * 2e-6,
    ((M + 63) // 64) * ((Cout + 63) // 64) * (64 + 64) * K * 2e-6))
