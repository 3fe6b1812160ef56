"""DynamicConv core alone at the decoder's shape (T = 32, B = 32, C = 1024, H = 16, DropConnect 0.1): device time of the
forward and backward launches (10 per hipGraph) and the achieved fraction of the HBM roofline on the ALGORITHMIC bytes
((2 C + H K) x 2 B per (t, b) row forward; x, dy, taps read + dx, dlogits written backward).
TELL_DYNCONV_LDS=0 runs the wave-per-row kernels (no LDS tile) for comparison."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import hip
T, B, C, H, p = 32, 32, 1024, 16, 0.1
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


mode = 'wave-per-row' if os.environ.get('TELL_DYNCONV_LDS') == '0' else 'LDS-tiled'
print('DynamicConv core, %s kernels, T=%d B=%d C=%d H=%d (bf16):' % (mode, T, B, C, H))
for K in (3, 7, 15, 31):
    x = torch.randn(T, B, C, device='cuda').bfloat16(); lg = torch.randn(T, B, H * K, device='cuda').bfloat16()
    y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x); dl = torch.empty_like(lg)
    taps = torch.empty(T * B * H, K, device='cuda')
    tf = timed(lambda: hip.call('tell_dynconv_fwd', x, lg, y, taps, T, B, H, K, C // H, p, 1, 2, 1))
    tb = timed(lambda: hip.call('tell_dynconv_bwd', x, dy, taps, dx, 0, dl, T, B, H, K, C // H, p, 1, 2, 1))
    bf = T * B * (2 * C + H * K) * 2                      # x read, y written, logits read
    bb = T * B * ((3 * C + H * K) * 2 + H * K * 4)        # x, dy read, dx written, dlogits written, fp32 taps read
    print('  K=%2d  fwd %5.1f us  %5.2f TB/s (%4.1f %% of 8 TB/s) | bwd %5.1f us  %5.2f TB/s (%4.1f %%)  [%.1f / %.1f MB]'
          % (K, tf, bf / tf * 1e-6, bf / tf * 1e-6 / 8 * 100, tb, bb / tb * 1e-6, bb / tb * 1e-6 / 8 * 100, bf / 1e6, bb / 1e6))

# ---- the conv-block core as one launch (tell_dynconv_block_fwd: GLU + tap logits + softmax + DropConnect + K-tap sum)
# against the three launches it replaces; achieved bandwidth on the FUSED algorithmic bytes: h1 [T*B, 2C] and W_tap
# [H*K, C] read, gl and y [T*B, C] and the fp32 taps [T*B*H, K] written
from tell_amd import ops
hip.require_gpu()
print('conv-block core, fused launch vs glu + tap-logit GEMM + dynconv forward (same shapes):')
for K in (3, 7, 15, 31):
    h1 = torch.randn(T * B, 2 * C, device='cuda').bfloat16(); wt = (torch.randn(H * K, C, device='cuda') * 0.05).bfloat16()
    gl = torch.empty(T * B, C, device='cuda', dtype=torch.bfloat16); y = torch.empty_like(gl)
    taps = torch.empty(T * B * H, K, device='cuda')
    lg = torch.empty(T * B, H * K, device='cuda', dtype=torch.bfloat16)

    def three():
        hip.call('tell_glu_fwd', h1, gl, T * B, C, 1)
        ops.gemm(gl, wt, out=lg)
        hip.call('tell_dynconv_fwd', gl, lg, y, taps, T, B, H, K, C // H, p, 1, 2, 1)
    assert hip.call_rc('tell_dynconv_block_fwd', h1, wt, gl, y, taps, T, B, H, K, p, 1, 2) == 0
    tfu = timed(lambda: hip.call('tell_dynconv_block_fwd', h1, wt, gl, y, taps, T, B, H, K, p, 1, 2))
    t3 = timed(three)
    by = T * B * 2 * C * 2 + H * K * C * 2 + 2 * T * B * C * 2 + T * B * H * K * 4
    print('  K=%2d  fused %5.1f us  %5.2f TB/s (%4.1f %% of 8 TB/s on %.1f MB)  |  three launches %5.1f us'
          % (K, tfu, by / tfu * 1e-6, by / tfu * 1e-6 / 8 * 100, by / 1e6, t3))
