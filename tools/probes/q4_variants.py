"""gemm_nt_q4_kernel: schedule variants / ablations / the ping-pong kernel INTERLEAVED inside one process (same box, same
thermal state: separate processes differ by +-5 % on this pool), and the kernel's own s_memtime stamps per output tile.
  python tools/probes/q4_variants.py [rounds]
Configurations are library options (tell_set_option, given here in their TELL_<KEY> spelling) read per launch (csrc/gemm_q4.hip, csrc/gemm.hip)."""
import os, statistics, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the ablations (TELL_Q4_ABL, TELL_Q4E_VAR) exist only in the probe build: PROBES=1 transform-and-tell_amd/csrc/build.sh
os.environ.setdefault('TELL_LIB', os.path.join(_ROOT, 'transform-and-tell_amd', 'csrc', 'libtell_hip_probes.so'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd import hip, ops
hip.require_gpu()
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
REP = 10
M = 32 * 512
SHAPES = (('qkv', 3072, 1024, 0), ('out', 1024, 1024, 0), ('fc1+gelu', 4096, 1024, 2), ('fc2', 1024, 4096, 0))
CONFIGS = [('pp2', {'TELL_GEMM_Q4': '0'}), ('q4e', {'TELL_GEMM_Q4': '1', 'TELL_GEMM_Q4E': '1', 'TELL_Q4_ABL': '0'})]
for v in os.environ.get('Q4_VARS', '0,1,2').split(','):
    CONFIGS.append(('q4 var %s' % v, {'TELL_GEMM_Q4': '1', 'TELL_GEMM_Q4E': '0', 'TELL_Q4_VAR': v, 'TELL_Q4_ABL': '0'}))
if os.environ.get('Q4_ABLS', '1') != '0':
    CONFIGS += [('q4 no epilogue', {'TELL_GEMM_Q4': '1', 'TELL_Q4_VAR': '0', 'TELL_Q4_ABL': '1'}),
                ('q4 no stores', {'TELL_GEMM_Q4': '1', 'TELL_Q4_VAR': '0', 'TELL_Q4_ABL': '2'})]


def setenv(env):
    hip.apply_env(env)


def graph_of(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), hip.bound_stream():
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    return g


def time_graph(g):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * REP)


data = {}
for name, N, K, act in SHAPES:
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    bias = torch.randn(N, device='cuda'); y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    data[name] = (a, w, bias, y, act)
graphs = {}
for cname, env in CONFIGS:                       # the environment is read at launch = at CAPTURE time
    setenv(env)
    for name, N, K, act in SHAPES:
        a, w, bias, y, act = data[name]
        graphs[(cname, name)] = graph_of(lambda: ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act))
times = {k: [] for k in graphs}
for r in range(ROUNDS):
    for k, g in graphs.items():
        times[k].append(time_graph(g))
print('median (min) us per launch over %d interleaved rounds; TFLOP/s from the median' % ROUNDS)
print('%-16s' % '' + ''.join('%-26s' % s[0] for s in SHAPES) + 'layer total')
for cname, _ in CONFIGS:
    row, tot = '%-16s' % cname, 0.0
    for name, N, K, act in SHAPES:
        t = times[(cname, name)]
        med = statistics.median(t)
        tot += med
        row += '%6.1f (%6.1f) %5.0f TF    ' % (med, min(t), 2.0 * M * N * K / med * 1e-6)
    print(row + '%6.1f us' % tot)

# ---- the kernel's own stamps (TELL_Q4_ABL=3): per output tile  t0 -> [tile top wait] ta -> [K loop] tb -> t1 -> [epilogue] t2
setenv({'TELL_GEMM_Q4': '1', 'TELL_GEMM_Q4E': '0', 'TELL_Q4_VAR': '0', 'TELL_Q4_ABL': '3'})
khz = hip.lib().tell_wall_clock_khz()
print('\ns_memtime stamps of wave 0 (shader clocks), mean over workgroups; per tile: top = tile-top setup + wait for K tile 0 '
      '(and the previous stores), loop = K loop, epi = epilogue issue, gap = end of epilogue -> next tile top')
for name, N, K, act in SHAPES:
    a, w, bias, y, act = data[name]
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device='cuda')
    for _ in range(2):
        ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act, aux=dbg)
    torch.cuda.synchronize()
    d = dbg.view(256, 8, 8).cpu().double()
    nt = int((d[0, :, 0] != 0).sum())
    out = []
    for t in range(nt):
        top = (d[:, t, 1] - d[:, t, 0]).mean(); loop = (d[:, t, 2] - d[:, t, 1]).mean()
        epi = (d[:, t, 4] - d[:, t, 3]).mean()
        gap = (d[:, t + 1, 0] - d[:, t, 4]).mean() if t + 1 < nt else float('nan')
        out.append('tile %d: top %6.0f loop %7.0f (%5.0f / K tile) epi %6.0f gap %5.0f' % (t, top, loop, loop / (K // 64), epi, gap))
    live = d[:, 0, 0] != 0
    span = (d[live, nt - 1, 4].max() - d[live, 0, 0].min())
    print('%-9s %d tiles per workgroup, first-in to last-out %8.0f clk\n   ' % (name, nt, span) + '\n   '.join(out))
hip.apply_env({'TELL_Q4_ABL': '0'})

# ---- gemm_nt_q4e_kernel probes (TELL_Q4E_VAR = index into PROBES of tools/gen_q4e_loop.py; act 0 shapes only): time + stamps
# t0 -> [setup (+ wait)] ta -> [drain of the previous tile (+ wait)] td -> [K loop with the deferred stores] tb -> t1
NPROBE = int(os.environ.get('Q4E_PROBES', '1'))
pg = {}
for v in range(NPROBE):
    setenv({'TELL_GEMM_Q4': '1', 'TELL_GEMM_Q4E': '1', 'TELL_Q4E_VAR': str(v), 'TELL_Q4_DYNAMIC': os.environ.get('Q4E_DYN', '1')})
    for name, N, K, act in SHAPES:
        if act == 0:
            a, w, bias, y, act = data[name]
            pg[(v, name)] = graph_of(lambda: ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act))
pt = {k: [] for k in pg}
for r in range(ROUNDS):
    for k, g_ in pg.items():
        pt[k].append(time_graph(g_))
print('\nq4e probes: median us per launch;  stamps in shader clocks, mean over workgroups, of the middle tile (qkv) / the only tile')
for v in range(NPROBE):
    hip.apply_env({'TELL_Q4E_VAR': str(v)})
    row = 'probe %d  ' % v
    for name, N, K, act in SHAPES:
        if act != 0:
            continue
        a, w, bias, y, act = data[name]
        dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device='cuda')
        for _ in range(2):
            ops.gemm(a, w, out=y, bias=bias, bias_mode=1, act=act, aux=dbg)
        torch.cuda.synchronize()
        d = dbg.view(256, 8, 8).cpu().double()
        nt = int((d[0, :, 0] != 0).sum())
        t = 1 if nt > 1 else 0
        top = (d[:, t, 1] - d[:, t, 0]).mean(); drain = (d[:, t, 2] - d[:, t, 1]).mean(); loop = (d[:, t, 3] - d[:, t, 2]).mean()
        gap = (d[:, t + 1, 0] - d[:, t, 4]).mean() if t + 1 < nt else float('nan')
        row += '%s %6.1f us [top %4.0f drain %5.0f loop %6.0f = %4.0f/Kt gap %5.0f]  ' % (name, statistics.median(pt[(v, name)]), top, drain, loop, loop / (K // 64), gap)
    print(row)
hip.apply_env({'TELL_Q4E_VAR': None})
