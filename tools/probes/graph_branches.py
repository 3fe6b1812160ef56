"""Do independent branches of a captured hipGraph (forked onto side streams during capture) run concurrently at replay?
Four chains of 40 small kernels each (one workgroup wave, ~10 us) captured (a) back to back on one stream, (b) on four
streams forked from / joined to the capturing stream."""
import sys, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import hip
M = 1024
xs = [torch.randn(M, M, device='cuda').bfloat16() for _ in range(4)]
ws = [torch.randn(M, M, device='cuda').bfloat16() for _ in range(4)]
ys = [[torch.empty(M, M, device='cuda', dtype=torch.bfloat16) for _ in range(2)] for _ in range(4)]


def chain(c):
    src = xs[c]
    for i in range(40):
        dst = ys[c][i & 1]
        hip.call('tell_gemm_nt', src, M, ws[c], M, dst, M, M, M, M, 1, 1, None, 0, 0, None, 1.0 / 32, 0, None)
        src = dst


def timed(g):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


for c in range(4):
    chain(c)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    for c in range(4):
        chain(c)
side = [torch.cuda.Stream() for _ in range(4)]
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    main = torch.cuda.current_stream()
    for c in range(4):
        side[c].wait_stream(main)
        with torch.cuda.stream(side[c]):
            chain(c)
    for c in range(4):
        main.wait_stream(side[c])
print('one stream: %.3f ms   four forked streams: %.3f ms' % (timed(g1), timed(g2)))
