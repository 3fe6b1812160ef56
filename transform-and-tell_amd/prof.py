"""Live per-kernel timing with HIP events on the launch stream (torch's current stream), used by
bench.py for the `roofline` object.  Disabled (zero overhead) unless `enable()` is called."""
import torch

_enabled = False
MIN_WORK = 2e9        # only launches with >= 2 GFLOP are timed (keeps the instrumentation out of the small kernels)
_records = []   # (kernel name, work, start event, end event)


def enable(flag=True):
    global _enabled
    _enabled = flag
    if flag:
        _records.clear()


def enabled():
    return _enabled


def begin(name, work):
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    return (name, work, ev0, ev1)


def end(rec):
    rec[3].record()
    _records.append(rec)


def summary():
    """-> {kernel: dict(launches, total_ms, avg_us, work)}; call after torch.cuda.synchronize()."""
    out = {}
    for name, work, e0, e1 in _records:
        d = out.setdefault(name, dict(launches=0, total_ms=0.0, work=0.0))
        d['launches'] += 1
        d['total_ms'] += e0.elapsed_time(e1)
        d['work'] += work
    for d in out.values():
        d['avg_us'] = 1e3 * d['total_ms'] / max(d['launches'], 1)
    return out
