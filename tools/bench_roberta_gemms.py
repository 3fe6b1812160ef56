"""Device time of RoBERTa-large's four projection GEMMs at B x 512 rows (bf16, bias epilogue, GELU on fc1), 10 launches per
hipGraph: which kernel the library picks (tell_gemm_nt_plan) and its TFLOP/s in isolation."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()        # (registers the tile-counter buffer of the resident GEMM launches)
REP = 10
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 512


def graph_of(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    return g


def time_graph(g, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * REP)


# all four graphs first, two untimed rounds (an idle GPU needs tens of milliseconds of load before its clocks settle: the
# first shape measured from cold read 15-20 % slow), then ROUNDS timed rounds over the shapes in turn; median per shape
ROUNDS = 7
cases = []
for name, N, K, act in (('qkv', 3072, 1024, 0), ('out', 1024, 1024, 0), ('fc1+gelu', 4096, 1024, 2), ('fc2', 1024, 4096, 0)):
    # (PITCH_PAD_A / _W / _C: extra elements per row of the activation / weight / output - probes for power-of-two row pitches)
    pa, pw, pc = (int(os.environ.get('PITCH_PAD_' + k, '0')) for k in 'AWC')
    a = torch.randn(M, K + pa, device='cuda').bfloat16()[:, :K]; w = torch.randn(N, K + pw, device='cuda').bfloat16()[:, :K]
    bias = torch.randn(N, device='cuda'); y = torch.empty(M, N + pc, device='cuda', dtype=torch.bfloat16)[:, :N]
    args = (a, a.stride(0), w, w.stride(0), y, y.stride(0), M, N, K, 1, 1, bias, 1, act, None, 1.0, 0, None)
    kname = hip.query('tell_gemm_nt_plan', *args)
    g = graph_of(lambda a=a, w=w, y=y, bias=bias, act=act: ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act))
    cases.append((name, N, K, kname, g, (a, w, bias, y), []))
for r in range(2 + ROUNDS):
    for c in cases:
        t = time_graph(c[4])
        if r >= 2:
            c[6].append(t)
tot_t = tot_f = 0.0
for name, N, K, kname, g, keep, ts in cases:
    t = statistics.median(ts)
    f = 2.0 * M * N * K
    tot_t += t; tot_f += f
    print('%-9s M=%d N=%4d K=%4d  %-34s %7.1f us  %6.0f TFLOP/s   (min %.1f max %.1f over %d rounds)'
          % (name, M, N, K, kname, t, f / t * 1e-6, min(ts), max(ts), ROUNDS))
print('layer total %.1f us -> %.0f TFLOP/s; x24 layers = %.2f ms' % (tot_t, tot_f / tot_t * 1e-6, 24 * tot_t * 1e-3))
