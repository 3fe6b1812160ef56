"""ctypes binding of libtell_hip.so (C ABI declared in include/tell_hip.h).

The prototypes are parsed from the header itself, so the binding cannot drift
from the ABI.  There is NO fallback: if the library is missing or a kernel
fails, a RuntimeError is raised - the product never silently runs on CPU.
"""
import ctypes
import os
import re

import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# TELL_LIB: another build of the same ABI (tools/probes/ load csrc/libtell_hip_probes.so, the -DTELL_PROBES build)
LIB_PATH = os.environ.get('TELL_LIB') or os.path.join(_HERE, 'csrc', 'libtell_hip.so')
HEADER_PATH = os.path.join(_ROOT, 'include', 'tell_hip.h')

F32, BF16 = 0, 1

_CTYPES = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float,
    'uint32_t': ctypes.c_uint32, 'uint64_t': ctypes.c_uint64, 'tell_stream_t': ctypes.c_void_p, 'void': None,
}


def _ctype(decl):
    decl = decl.strip()
    if decl == 'const char*':
        return ctypes.c_char_p
    if '*' in decl:
        return ctypes.c_void_p
    return _CTYPES[decl.replace('const ', '')]


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every `tell_*` prototype."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(tell_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                mm = re.match(r'(.*?)(\w+)$', a)
                argtypes.append(_ctype(mm.group(1)))
                argnames.append(mm.group(2))
        protos[name] = (_ctype(ret), argtypes, argnames)
    return protos


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'tell_amd: %s not found - build it with `python __graft_entry__.py build` '
                '(or transform-and-tell_amd/csrc/build.sh). There is no CPU fallback.' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (res, argtypes, _) in _protos.items():
            fn = getattr(_lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = argtypes
        _options_from_env()
    return _lib


# ---- run-time options of the library (include/tell_hip.h tell_set_option; keys and defaults in csrc/options.h).
# The library itself reads no environment variable.  For the A/B scripts under tools/ the host mirror translates
# TELL_<KEY> (e.g. TELL_GEMM_Q4=0) into tell_set_option calls ONCE, when the library is loaded; inside a process use
# set_option / `with hip.options(gemm_s64=0):`.
def option_keys():
    out, i = [], 0
    while True:
        k = lib().tell_option_key(i)
        if k is None:
            return out
        out.append(k.decode())
        i += 1


def option_defaults():
    return {k: lib().tell_option_default(i) for i, k in enumerate(option_keys())}


def set_option(key, value):
    if lib().tell_set_option(key.encode(), int(value)) != 0:
        raise KeyError('tell_amd: %s' % _lib.tell_last_error().decode())


def get_option(key):
    v = lib().tell_get_option(key.encode())
    if v == -(1 << 63):
        raise KeyError('tell_amd: unknown option %r' % key)
    return v


class options:
    """`with hip.options(gemm_s64=0, conv_tile=2):` - set for the block, previous values restored after it."""

    def __init__(self, **kw):
        self.kw, self.prev = kw, {}

    def __enter__(self):
        lib()
        for k, v in self.kw.items():
            self.prev[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False


def apply_env(env):
    """{'TELL_GEMM_Q4': '0', ...} (the spelling the A/B scripts under tools/ use) -> set_option calls; a value of None
    restores the option's load-time value.  Names that are not library options are ignored (host-side switches)."""
    lib()
    for name, v in env.items():
        k = name[5:].lower() if name.startswith('TELL_') else name
        if k in _loaded_options:
            set_option(k, _loaded_options[k] if v is None else int(v))


_loaded_options = {}


def loaded_options():
    """{key: value} as they stood right after the library was loaded (defaults + what TELL_<KEY> asked for)."""
    lib()
    return dict(_loaded_options)


def _options_from_env():
    i = 0
    while True:
        k = _lib.tell_option_key(i)
        if k is None:
            return
        v = os.environ.get('TELL_' + k.decode().upper())
        if v is not None and v.strip() != '':
            _lib.tell_set_option(k, int(v))
        _loaded_options[k.decode()] = _lib.tell_get_option(k)
        i += 1


_tile_queue = {}


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('tell_amd: no HIP device visible (torch.cuda.is_available() is False); '
                           'the MI355X kernels have no CPU fallback')
    lib()
    dev = torch.cuda.current_device()
    if dev not in _tile_queue:            # work-queue counters of the persistent 256x256 GEMM launches (csrc/gemm.hip)
        # 4 MB per device: 8 counters per launch; launches recorded into hipGraphs keep theirs (first half: 65536 launches'
        # worth) until the graph's owner gives them back (tile_slots below)
        _tile_queue[dev] = torch.zeros(1 << 20, dtype=torch.int32, device='cuda')
        lib().tell_gemm_set_tile_queue(_tile_queue[dev].data_ptr(), 1 << 20, None)


class tile_slots:
    """`with hip.tile_slots() as held:` around a stream capture: records which tile-counter slots the captured resident
    GEMM launches took (csrc/gemm.hip) - keep `held` next to the CUDAGraph object; when it is dropped (graph evicted,
    cache cleared) the slots go back to the device's free list.  A graph that is destroyed without this simply leaks its
    slots (65536 per device; later captures then walk static tile lists)."""

    def __init__(self):
        self.slots, self.device = None, None

    def __enter__(self):
        self.device = torch.cuda.current_device()
        lib().tell_gemm_tile_queue_log_begin()
        return self

    def __exit__(self, *exc):
        cap = max(int(lib().tell_gemm_tile_queue_log_count()), 1)      # (every slot: a truncated list would leak the rest)
        buf = (ctypes.c_int * cap)()
        n = lib().tell_gemm_tile_queue_log_end(buf, cap)
        self.slots = list(buf[:min(n, cap)])
        return False

    def release(self):
        if self.slots and _lib is not None:
            try:
                arr = (ctypes.c_int * len(self.slots))(*self.slots)
                if torch.cuda.current_device() == self.device:
                    _lib.tell_gemm_tile_queue_release(arr, len(self.slots))
                else:
                    with torch.cuda.device(self.device):
                        _lib.tell_gemm_tile_queue_release(arr, len(self.slots))
            except Exception:              # noqa: BLE001 - interpreter shutdown
                pass
        self.slots = None

    def __del__(self):
        self.release()


def tile_queue_stats():
    """(slots, fresh captured slots handed out, free-list length) of the current device's tile-counter buffer."""
    out = (ctypes.c_int * 3)()
    lib().tell_gemm_tile_queue_stats(out)
    return tuple(out)


def dt(t):
    """dtype code of a tensor / torch dtype."""
    d = t.dtype if isinstance(t, torch.Tensor) else t
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise TypeError('tell_amd: unsupported dtype %s (float32 / bfloat16 only)' % d)


def _conv(a):
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


_bound_stream = None      # raw hipStream_t bound for a whole step (avoids a torch lookup per launch)
_fns = {}


class bound_stream:
    """`with hip.bound_stream():` pins torch's CURRENT stream for every launch inside the block
    (one lookup per training step instead of one per kernel)."""

    def __enter__(self):
        global _bound_stream
        self.prev = _bound_stream
        _bound_stream = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        global _bound_stream
        _bound_stream = self.prev
        return False


def query(name, *args):
    """Call an entry point that returns a string (tell_*_plan): tensors -> device pointers, stream = NULL."""
    fn = getattr(lib(), name)
    res = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], None)
    return res.decode() if res is not None else ''


def call_rc(name, *args):
    """call() for the entry points that may DECLINE a shape: -> the positive return code (0 = launched, 1 = not a shape
    this kernel takes, nothing launched); errors still raise."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    stream = _bound_stream if _bound_stream is not None else torch.cuda.current_stream().cuda_stream
    rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], stream)
    if rc < 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, lib().tell_last_error().decode()))
    return rc


def call(name, *args):
    """Call a C-ABI entry point; tensors -> device pointers; last arg (stream) added here."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    stream = _bound_stream if _bound_stream is not None else torch.cuda.current_stream().cuda_stream
    rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], stream)
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, lib().tell_last_error().decode()))
