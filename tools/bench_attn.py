"""RoBERTa self-attention forward alone (B=16, H=16, S=512, D=64, bf16, key-padding mask, dropout 0.1 / 0)."""
import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import hip
from tell_amd import runtime as rt
B, H, S, D = 16, 16, 512, 64
E = H * D
qkv = torch.randn(B * S, 3 * E, device='cuda').bfloat16()
out = torch.empty(B * S, E, device='cuda', dtype=torch.bfloat16)
mask = torch.zeros(B, S, dtype=torch.uint8, device='cuda')
mask[:, 480:] = 1


def run(p, m):
    hip.call('tell_attn_fwd', qkv, qkv[:, E:], qkv[:, 2 * E:], out, None, m, None, None, B, H, S, S, D,
             3 * E, S * 3 * E, 3 * E, S * 3 * E, 3 * E, S * 3 * E, E, S * E, 0, p, rt.seed(), 7, hip.BF16)


for p, m, name in ((0.1, mask, 'dropout 0.1 + mask'), (0.1, None, 'dropout 0.1'), (0.0, None, 'no dropout')):
    for _ in range(5):
        run(p, m)
    torch.cuda.synchronize()
    ts = []
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run(p, m)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 30)
    us = sorted(ts)[1]
    print('%-20s %7.1f us  %6.1f TFLOP/s' % (name, us, 4.0 * B * H * S * S * D / us / 1e6))
