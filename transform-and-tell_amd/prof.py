"""Live per-kernel timing with HIP events on the launch stream (torch's current stream), used by
bench.py for the `roofline` object.  Disabled (zero overhead) unless `enable()` is called."""
import torch

_enabled = False
MIN_WORK = 2e9        # only launches with >= 2 GFLOP are timed (keeps the instrumentation out of the small kernels)
SAMPLE = 3           # time every 3rd qualifying launch of a kernel (timed events fence the stream, keep them
                     # sparse; 3 is coprime to the 4-GEMM period of a RoBERTa layer, so every shape is sampled)
_records = []   # (kernel name, work, start event, end event)
_seen = {}


def enable(flag=True):
    global _enabled
    _enabled = flag
    if flag:
        _records.clear()
        _seen.clear()


def reset_records():
    """Forget the event-bracketed samples so far (the stamps inside captured graphs stay: they are part of the graphs)."""
    _records.clear()
    _seen.clear()


# ---- launches inside captured hipGraphs: the kernel records its own execution span (csrc/gemm.hip gemm_ts_*) --------
GRAPH_SAMPLE = 4          # every 4th qualifying launch records its execution span
_g = {'buf': None, 'n': 0, 'slots': [], 'seen': {}, 'khz': 0}


def graph_begin(name, work):
    """Called while a stream is capturing: -> slot index if this launch is sampled (the caller launches the kernel and
    then calls graph_end(slot)), else None."""
    from . import hip
    n = _g['seen'].get(name, 0)
    _g['seen'][name] = n + 1
    if n % GRAPH_SAMPLE:
        return None
    if _g['buf'] is None:
        _g['buf'] = torch.zeros(4 * 4096, dtype=torch.int64, device='cuda')      # 4 words per slot: in, out, arrivals, -
        _g['khz'] = hip.lib().tell_wall_clock_khz()
    i = _g['n']
    if i >= 4096:
        return None
    _g['n'] = i + 1
    _g['slots'].append((name, work))
    # the kernel itself records [first workgroup in, last workgroup out] (csrc/gemm.hip gemm_ts_enter / gemm_ts_exit):
    # the interval an external profiler reports as the kernel's duration, without the dispatch waits a bracket of
    # neighbouring launches would add when other streams keep the CUs busy
    hip.call('tell_gemm_ts_next', _g['buf'][4 * i:])
    return i


def graph_end(i):
    pass


def graph_summary():
    """-> {kernel: dict(timed, total_ms, avg_us, work)} from the stamps of the LAST replay of every captured graph;
    call after torch.cuda.synchronize()."""
    out = {}
    if _g['buf'] is None or not _g['khz']:
        return out
    v = _g['buf'][:4 * _g['n']].tolist()
    for i, (name, work) in enumerate(_g['slots']):
        t0, t1 = v[4 * i], v[4 * i + 1]
        if t0 == 0 or t1 <= t0:
            continue                                   # a graph that was captured but not replayed since
        if t0 < 0 or t1 < 0:
            continue                                   # (~0 as int64): armed but the kernel had no timestamp hooks
        ms = max(t1 - t0, 1.0) / _g['khz']
        d = out.setdefault(name, dict(timed=0, total_ms=0.0, work=0.0))
        d['timed'] += 1
        d['total_ms'] += ms
        d['work'] += work
    for d in out.values():
        d['avg_us'] = 1e3 * d['total_ms'] / max(d['timed'], 1)
    return out


def enabled():
    return _enabled


def sampled(name):
    """True for the launches of `name` that get an event bracket (1 in SAMPLE)."""
    n = _seen.get(name, 0)
    _seen[name] = n + 1
    return n % SAMPLE == 0


def begin(name, work):
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    return (name, work, ev0, ev1)


def end(rec):
    rec[3].record()
    _records.append(rec)


_overhead_ms = [0.0]


TINY_KERNEL_US = 1.5      # execution time assumed for the 1-element kernel used by calibrate()


def calibrate(n=40):
    """What an event bracket adds to the kernel inside it: the two timing packets AND the dispatch latency of a
    kernel that follows a timing packet (without events the command processor overlaps that dispatch with the
    tail of the previous kernel, which is what rocprofv3's kernel timestamps see).  Measured as the bracket
    around a 1-element kernel minus its execution time; subtracted from every measured bracket."""
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    x = torch.zeros(1 << 20, device='cuda')
    y = torch.zeros(1, device='cuda')
    for e0, e1 in evs:
        x.add_(1)                  # real work in front, as in a training step
        e0.record()
        y.add_(1)
        e1.record()
    torch.cuda.synchronize()
    d = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    _overhead_ms[0] = max(d[len(d) // 2] - TINY_KERNEL_US * 1e-3, 0.0)
    return _overhead_ms[0] * 1e3


def overhead_us():
    return _overhead_ms[0] * 1e3


def summary():
    """-> {kernel: dict(launches, timed, total_ms, avg_us, work)}; call after torch.cuda.synchronize()."""
    out = {}
    for name, work, e0, e1 in _records:
        d = out.setdefault(name, dict(timed=0, total_ms=0.0, work=0.0))
        d['timed'] += 1
        d['total_ms'] += max(e0.elapsed_time(e1) - _overhead_ms[0], 1e-4)
        d['work'] += work                                   # work of the TIMED launches only
    for name, d in out.items():
        d['launches'] = _seen.get(name, d['timed'])         # all qualifying launches, timed or not
        d['avg_us'] = 1e3 * d['total_ms'] / max(d['timed'], 1)
    return out
