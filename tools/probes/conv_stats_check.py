"""tell_conv_bn_stats against torch (fp32 conv on the bf16-rounded inputs): raw output, batch mean, invstd; repeated
launches on the SAME workspace with different data and shapes (stale partials would show up)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip
torch.manual_seed(0)
ws = torch.empty(1 << 22, dtype=torch.float32, device='cuda')
zero = torch.zeros(256, dtype=torch.uint8, device='cuda')
cases = [(2, 14, 14, 256, 3, 1, 256), (32, 14, 14, 256, 3, 1, 256), (32, 14, 14, 1024, 1, 1, 256), (32, 14, 14, 256, 1, 1, 1024),
         (8, 56, 56, 64, 3, 1, 64), (8, 56, 56, 256, 1, 2, 512), (8, 28, 28, 128, 3, 2, 128), (3, 7, 9, 512, 3, 1, 512)]
for rep in range(3):
    for (B, H, W, Cin, k, s, Cout) in cases:
        for tile in ('1', '2', '3'):
            hip.apply_env({'TELL_CONV_TILE': tile})
            p = k // 2
            OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            x = (torch.randn(B, H, W, Cin, device='cuda') + 0.3 * rep).bfloat16()
            w = (torch.randn(Cout, k, k, Cin, device='cuda') * 0.05).bfloat16()
            y = torch.empty(B * OH * OW, Cout, dtype=torch.bfloat16, device='cuda')
            mean = torch.empty(Cout, device='cuda'); invstd = torch.empty(Cout, device='cuda')
            rm = torch.zeros(Cout, device='cuda'); rv = torch.ones(Cout, device='cuda')
            hip.call('tell_conv_bn_stats', x, w.reshape(Cout, -1), y, B, H, W, Cin, k, k, s, p, OH, OW, Cout, 1e-5, 0.1,
                     mean, invstd, rm, rv, ws, zero)
            torch.cuda.synchronize()
            ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=s, padding=p)
            ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
            yr = ref.bfloat16().float()
            e_y = ((y.float() - yr).norm() / yr.norm()).item()
            m_ref = y.float().mean(0); v_ref = y.float().var(0, unbiased=False)
            e_m = ((mean - m_ref).abs().max() / (m_ref.abs().max() + 1e-6)).item()
            e_v = ((invstd - torch.rsqrt(v_ref + 1e-5)).abs().max() / torch.rsqrt(v_ref + 1e-5).abs().max()).item()
            bad = e_y > 1e-2 or e_m > 1e-3 or e_v > 1e-3
            print('rep %d B%d %dx%d Cin%d k%d s%d Cout%d tile %s: y %.2e mean %.2e invstd %.2e %s'
                  % (rep, B, H, W, Cin, k, s, Cout, tile, e_y, e_m, e_v, 'BAD' if bad else ''), flush=True)
