"""1024-row GEMMs with a long reduction (decoder fc2 / context_fc forward, fc1 input gradient: [1024, 1024] from K = 4096;
linear1's input gradient K = 2048) - the single launch the library picks against K slices of one grouped launch + the
fold, per tile size of the grouped kernel (TELL_GROUP_TILE=64 / 128 is read once per process)."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()
REP = 10


def timed(fn):
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


print('TELL_GROUP_TILE =', os.environ.get('TELL_GROUP_TILE', '(default)'))
for M, N, K in ((1024, 1024, 4096), (1024, 4096, 1024), (1024, 1024, 2048), (1024, 2048, 1024)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda'); y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    t0 = timed(lambda: ops.gemm(a, w, out=y, bias=bias, bias_mode=1))
    ref = y.clone()
    line = 'M=%d N=%d K=%d  single launch %5.1f us' % (M, N, K, t0)
    for splits in (2, 4, 8):
        ks = K // splits
        if ks % 64:
            continue
        partial = torch.empty(splits, M, N, dtype=torch.float32, device='cuda')

        def split():
            ops.gemm_grouped([dict(a=a[:, i * ks:(i + 1) * ks], b=w[:, i * ks:(i + 1) * ks], out=partial[i], form='nt')
                              for i in range(splits)])
            hip.call('tell_splitk_reduce', partial, splits, partial.stride(0), M, N, bias, 0, 1.0, y, y.stride(0), hip.BF16)
        t = timed(split)
        err = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
        line += ' | %d slices %5.1f us (err %.1e)' % (splits, t, err)
    print(line)
