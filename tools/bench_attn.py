import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import hip, runtime as rt
B, H, S, E = 16, 16, 512, 1024
qkv = (torch.randn(B * S, 3 * E, device='cuda') * 0.5).bfloat16()
out = torch.empty(B * S, E, device='cuda', dtype=torch.bfloat16)
mask = torch.zeros(B, S, dtype=torch.uint8, device='cuda')
def run(p):
    hip.call('tell_attn_fwd', qkv, qkv[:, E:], qkv[:, 2 * E:], out, None, mask, None, None, B, H, S, S, 64, 3 * E, S * 3 * E,
             3 * E, S * 3 * E, 3 * E, S * 3 * E, E, S * E, 0, p, 1, 2, 1)
for p in (0.0, 0.1):
    for _ in range(3): run(p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(p)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('roberta self-attn B16 H16 S512 D64 p=%.1f: %.1f us  %.1f TF/s' % (p, us, 4.0 * B * H * S * S * 64 / us / 1e6))
