"""hipGraph replay of fixed-shape kernel sequences.

The frozen encoders are ~650 small launches per step; issued one by one from Python they cost
more host time than GPU time.  `GraphedCall` runs the function eagerly once (builds weight caches),
records it into a hipGraph right after that first call and replays the graph from the second call
on: one launch per step.  Kernel arguments are frozen at capture; dropout stays fresh through a device step
counter (`rng=True`, csrc/common.h tell_step_salt); inputs are copied into the captured static buffers.

TELL_GRAPHS=0 disables capture; a failed capture falls back to eager execution for that signature."""
import os

import torch

from . import hip


class no_gc:
    """`with graphs.no_gc():` around a stream capture.  Python's cyclic collector may run at any allocation; if it
    frees an object that owns a HIP resource (an old CUDAGraph, an event, a stream held by a dropped trainer) while a
    capture is open, the destructor's hipGraphDestroy / hipEventDestroy is illegal there and the process ABORTS
    ("Fatal Python error: Aborted ... Garbage-collecting" - seen once in ~30 test-suite runs).  Collect before, keep the
    collector off inside, restore after."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False

ENABLED = os.environ.get('TELL_GRAPHS', '1') != '0'


MAX_SIGNATURES = 8      # every captured signature keeps its activations in a private pool (2-4 GB at B = 32; 288 GB HBM)
# A signature is captured at its CAPTURE_AFTER-th sighting: real BucketIterator batches pad to the per-batch maximum,
# so most shapes are seen once - capturing each of them would cost a host capture pass and pin a private activation
# pool per shape for nothing.  Fixed-shape runs (bench.py) pass capture_after=1.
CAPTURE_AFTER = max(1, int(os.environ.get('TELL_GRAPH_AFTER', '2')))
MAX_ENTRIES = 256       # bookkeeping entries (sighting counters) kept per cache; the oldest are forgotten beyond that
THRASH_WINDOW, THRASH_EVICTIONS, THRASH_FREEZE = 64, 8, 256          # (see SignatureCache)


class SignatureCache:
    """signature -> entry dict, in least-recently-used order.  Entries start as {'state': 'seen', 'hits': 0}; the owner
    turns one into 'ready' (with its graph objects) after a capture.  make_room() evicts the least recently used READY
    entries down to max_ready - 1 (dropping the graph releases its private pool) and forgets the oldest counters
    beyond MAX_ENTRIES: neither the captured pools nor the dict grow with the number of distinct shapes seen."""

    def __init__(self, max_ready, capture_after=None):
        from collections import OrderedDict
        self.entries = OrderedDict()
        self.max_ready = max(1, int(max_ready))
        self.capture_after = CAPTURE_AFTER if capture_after is None else max(1, int(capture_after))
        self.evictions = 0
        # thrash guard: more distinct signatures in rotation than max_ready makes an LRU evict a graph at almost every
        # miss - each eviction is a device synchronisation, a destroyed pool and an ~80 ms capture pass for a graph that
        # is gone again before it has paid that back (worse than never capturing).  More than THRASH_EVICTIONS evictions
        # within THRASH_WINDOW sightings freeze the set for THRASH_FREEZE sightings: what is captured keeps replaying,
        # everything else runs eagerly; then the LRU gets another chance (the mix of shapes may have changed).
        self.clock = 0
        self.recent_evictions = []
        self.frozen_until = -1
        self.freezes = 0

    def touch(self, sig):
        """-> the entry of sig (created on first sight), counted as one more sighting and made most recently used."""
        e = self.entries.get(sig)
        if e is None:
            e = self.entries[sig] = {'state': 'seen', 'hits': 0}
        e['hits'] += 1
        self.clock += 1
        self.entries.move_to_end(sig)
        return e

    def due(self, e):
        return e['state'] == 'seen' and e['hits'] >= self.capture_after

    def make_room(self):
        """Call right before a capture (never inside one: destroying a graph while capturing is illegal).
        -> True: there is room for one more capture; False: the set is full and frozen (thrash guard) - do not capture."""
        ready = [k for k, v in self.entries.items() if v['state'] == 'ready']
        if len(ready) >= self.max_ready:
            if self.clock < self.frozen_until:
                return False
            self.recent_evictions = [c for c in self.recent_evictions if c > self.clock - THRASH_WINDOW]
            if len(self.recent_evictions) >= THRASH_EVICTIONS:
                self.frozen_until = self.clock + THRASH_FREEZE
                self.freezes += 1
                self.recent_evictions = []
                return False
            if torch.cuda.is_available():
                torch.cuda.synchronize()            # the victim's last replay may still be running
        while len(ready) >= self.max_ready:
            k = ready.pop(0)
            self.entries.pop(k).clear()             # drops the graph, its static buffers and its pool
            self.evictions += 1
            self.recent_evictions.append(self.clock)
        if len(self.entries) > MAX_ENTRIES:
            for k in [k for k, v in self.entries.items() if v['state'] != 'ready'][:len(self.entries) - MAX_ENTRIES]:
                del self.entries[k]
        return True

    def clear(self):
        self.entries.clear()


class GraphedCall:
    def __init__(self, fn, name='graph', rng=False, buffers=2, capture_after=None):
        """rng: the function contains dropout.  Its kernels are captured with a device step counter registered
        (tell_set_rng_step_ptr) and the counter is bumped before every replay, so the frozen seed/salt arguments
        still give fresh masks (csrc/common.h tell_step_salt).

        buffers: a replay writes its result into buffers owned by the graph, so every signature is captured
        `buffers` times and the captures are used round-robin: the tensor returned by a call stays intact until
        `buffers` further calls - the step pipeline needs 2 (the encoders of batch N+1 are replayed while the
        decoder step of batch N is still reading the outputs for batch N).

        capture_after: sightings of a signature before it is captured (default graphs.CAPTURE_AFTER); at most
        MAX_SIGNATURES signatures stay captured, least recently used evicted first."""
        self.fn = fn
        self.name = name
        self.rng = rng
        self.buffers = max(1, int(os.environ.get('TELL_GRAPH_BUFFERS', buffers)))   # env: test aid
        self.cache = SignatureCache(MAX_SIGNATURES, capture_after)
        self.entries = self.cache.entries  # signature -> dict(state=..., slots=[dict(graph, static_in, static_out, counter)], turn)

    def reset(self):
        self.cache.clear()

    def __call__(self, x, key=(), slot=None):
        """x: the single tensor input; key: extra hashable state the kernel sequence depends on.  slot: which of the
        signature's `buffers` captures to replay (modulo) instead of this signature's own round-robin turn - callers that
        run several GraphedCalls per step pass ONE shared counter, so the buffer addresses a consumer sees depend on the
        step parity alone and not on how often each signature has been seen (training/step_graph.py keys its captures on
        those addresses: with per-signature turns, 3 article lengths x 2 RoBERTa slots x 2 ResNet slots were 12 step
        graphs, each an eager step and a capture away)."""
        self.last_replayed = False          # True: the result just returned lives in a graph-owned static buffer
        if not ENABLED or not x.is_cuda:
            return self.fn(x)
        sig = (tuple(x.shape), x.dtype, x.device.index, key)
        e = self.cache.touch(sig)
        if e['state'] == 'seen':
            out = self.fn(x)                            # eager: also builds the weight caches the capture relies on
            if self.cache.due(e) and self.cache.make_room():
                self._capture(e, x)                     # records, does not execute: this call pays for it,
            return out                                  # not a later (timed) one
        if e['state'] != 'ready':
            return self.fn(x)
        if slot is None:
            s = e['slots'][e['turn']]
            e['turn'] = (e['turn'] + 1) % len(e['slots'])
        else:
            s = e['slots'][slot % len(e['slots'])]
        s['generation'] = s.get('generation', 0) + 1       # consumers holding this slot's output can detect reuse
        self.last_slot = s
        s['static_in'].copy_(x)
        if s['counter'] is not None:
            s['counter'].fill_(e['replays'])            # same stream as the replay: ordered before it
        e['replays'] += 1
        s['graph'].replay()
        self.last_replayed = True
        return s['static_out']

    def _capture(self, e, x):
        try:
            slots = []
            for _ in range(self.buffers):
                static_in = x.clone()
                counter = torch.zeros(1, dtype=torch.int32, device=x.device) if self.rng else None
                g = torch.cuda.CUDAGraph()
                held = hip.tile_slots()                 # tile-counter slots of the captured resident GEMMs: back to the
                try:                                    # free list when this entry is evicted (held.__del__)
                    if counter is not None:
                        hip.call('tell_set_rng_step_ptr', counter)
                    # thread_local: calls made by OTHER threads while we capture (the RCCL watchdog of a data-parallel
                    # run polls events) must not invalidate the capture; everything captured is issued from this thread
                    with no_gc(), held, torch.cuda.graph(g, capture_error_mode='thread_local'):
                        with hip.bound_stream():        # launches must go to the CAPTURING stream
                            static_out = self.fn(static_in)
                finally:
                    if counter is not None:
                        hip.call('tell_set_rng_step_ptr', None)
                slots.append({'graph': g, 'static_in': static_in, 'static_out': static_out, 'counter': counter,
                              'tile_slots': held})
            e.update(state='ready', slots=slots, turn=0, replays=1)
        except Exception as exc:                        # noqa: BLE001 - any capture problem -> eager for good
            e['state'] = 'failed'
            e['error'] = repr(exc)
