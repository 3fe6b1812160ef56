"""Attention kernels alone at the shapes of the step (bf16): RoBERTa self-attention (B = 32, H = 16, S = 512, D = 64,
attention dropout 0.1 as the reference leaves the hub model in train mode) forward, and the decoder's four context
attentions forward + backward (T = 32 queries; article 512 keys, image 49, faces 4, objects 64).  Reports device time
per launch (10 launches per hipGraph) and the MFMA rate on 4 T S D flops per (b, h) forward."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import hip
hip.require_gpu()        # (registers the attention mask scratch)
REP = 10


def timed(fn):
    """Median of 5 timed batches of 5 x 10 launches, after 2 untimed ones (an idle GPU's clocks settle over tens of
    milliseconds of load: the first kernel measured from cold read 10-15 % slow)."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3 / (5 * REP))
    return statistics.median(ts)


def case(name, B, H, T, S, D, p, bias, backward):
    E = H * D
    q = torch.randn(T, B, E, device='cuda').bfloat16()
    k = torch.randn(S, B, E, device='cuda').bfloat16(); v = torch.randn(S, B, E, device='cuda').bfloat16()
    out = torch.empty_like(q); lse = torch.empty(B * H, T, device='cuda')
    bk = torch.randn(E, device='cuda').bfloat16() if bias else None
    bv = torch.randn(E, device='cuda').bfloat16() if bias else None
    mask = torch.zeros(B, S, dtype=torch.uint8, device='cuda')
    has_zero = 1 if bias else 0

    def fwd():
        hip.call('tell_attn_fwd', q, k, v, out, lse, mask, bk, bv, B, H, T, S, D, q.stride(0), q.stride(1), k.stride(0),
                 k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1), has_zero, p, 1, 2, hip.dt(q))
    tf = timed(fwd)
    fl = 4.0 * T * (S + 2 * has_zero) * D * B * H
    line = '%-30s T=%3d S=%3d p=%.1f | fwd %6.1f us %5.0f TF' % (name, T, S, p, tf, fl / tf * 1e-6)
    if backward:
        dout = torch.randn_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
        dbk = torch.empty(B, E, device='cuda') if bias else None
        dbv = torch.empty(B, E, device='cuda') if bias else None

        def bwd():
            hip.call('tell_attn_bwd', q, k, v, out, dout, lse, mask, bk, bv, dq, dk, dv, dbk, dbv, B, H, T, S, D,
                     q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0),
                     out.stride(1), has_zero, p, 1, 2, hip.dt(q))
        tb = timed(bwd)
        line += ' | bwd %6.1f us %5.0f TF' % (tb, 2.5 * fl / tb * 1e-6)
    print(line)


case('roberta self', 32, 16, 512, 512, 64, 0.1, False, False)
case('roberta self (no drop)', 32, 16, 512, 512, 64, 0.0, False, False)
for name, S in (('decoder article', 512), ('decoder image', 49), ('decoder faces', 4), ('decoder objects', 64)):
    case(name, 32, 16, 32, S, 64, 0.1, True, True)
