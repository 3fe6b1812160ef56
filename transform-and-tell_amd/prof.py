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
