"""What a skinny linear pays for COLD weights: the same launch over 1 / 8 / 64 / 256 copies of its weights in rotation
(L2-hot / L2-resident across the XCDs / Infinity-Cache-resident / HBM), 256 launches per graph replay.
(Round 6 also measured every launch warming its successor's weights - csrc/decode.hip has the numbers; it lost.)
usage (GPU box): python tools/probes/skinny_cold.py [M]"""
import sys

import torch

sys.path.insert(0, '.')
import tell_amd  # noqa: E402
from tell_amd import decode, hip  # noqa: E402

hip.require_gpu()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
E, F = 1024, 4096
bf = dict(dtype=torch.bfloat16, device='cuda')
f32 = dict(dtype=torch.float32, device='cuda')
x, x4 = torch.randn(M, E, **bf), torch.randn(M, F, **bf)
o32, h = torch.empty(M, E, **f32), torch.empty(M, F, **bf)
LAUNCHES = 256


def timeit(make, copies, warm=False):
    made = [make() for _ in range(copies)]
    fns = [(lambda f=f: f(None)) for f, _ in made]
    for f in fns[:4]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with hip.bound_stream():
            for i in range(LAUNCHES):
                fns[i % copies]()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (4 * LAUNCHES)


def linear2():
    w, b = torch.randn(E, E, **bf) * 0.03, torch.randn(E, **f32)
    return (lambda nx: decode._skinny([x], E, [w], [b], [o32], E, M, E, E, res=x, ld_res=E, out_f32=True)), ([w], M, E, E, 0)


def fc1():
    w, b = torch.randn(F, E, **bf) * 0.03, torch.randn(F, **f32)
    return (lambda nx: decode._skinny([x], E, [w], [b], [h], F, M, F, E, act=1)), ([w], M, F, E, 1)


def fc2():
    w, b = torch.randn(E, F, **bf) * 0.03, torch.randn(E, **f32)
    return (lambda nx: decode._skinny([x4], F, [w], [b], [o32], E, M, E, F, res=x, ld_res=E, out_f32=True)), ([w], M, E, F, 0)


print('M = %d; us per launch by copies of the weights in rotation' % M)
for name, make, mb in (('linear2 N1024 K1024 (2 MB)', linear2, 2), ('fc1 N4096 K1024 (8 MB)', fc1, 8), ('fc2 N1024 K4096 (8 MB)', fc2, 8)):
    row = []
    for copies in (1, 8, 64, 256 if mb == 2 else 96):
        row.append('%d: %.2f' % (copies, timeit(make, copies)))
    print('%-28s %s' % (name, '   '.join(row)))
