"""Un-profiled GPU timeline of the pipelined training step: HIP timing events at the phase boundaries of every stream,
printed as offsets (ms) from the step's first event.  Shows what really overlaps (rocprofv3 serialises the queues)."""
import sys, time, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import hip, ops
from tell_amd import runtime as rt
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
from tell_amd.build import build_model
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
tr = Trainer(model, device='cuda', capture_after=1)
batches = [synthetic_batch(16, 512, 33, False, seed=1234 + i, device='cuda') for i in range(4)]
marks = []


def ev(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    marks.append((name, e, time.perf_counter()))


def wrap(obj, attr, name):
    f = getattr(obj, attr)

    def g(*a, **k):
        ev(name + '_start')
        r = f(*a, **k)
        ev(name + '_end')
        return r
    setattr(obj, attr, g)


wrap(model, '_run_roberta', 'roberta')
wrap(model, '_run_resnet', 'resnet')
wrap(tr, '_update', 'update')
fwd = model.forward


def fwd_marked(*a, **k):
    ev('decoder_fwd_start')
    r = fwd(*a, **k)
    ev('decoder_fwd_end')
    return r


model.forward = fwd_marked
jw = ops.join_wgrad_stream


def jw_marked():
    ev('backward_end(main)')
    jw()
    ev('wgrad_joined(main)')


ops.join_wgrad_stream = jw_marked
import tell_amd.training.trainer as T
T.ops.join_wgrad_stream = jw_marked


def fresh(b):
    return {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}


N = 10
steps = []
for i in range(N):
    marks = []
    ev('step_start')
    tr.train_one_batch(fresh(batches[i % 4]), next_batch=fresh(batches[(i + 1) % 4]))
    steps.append(marks)
torch.cuda.synchronize()
prev0 = None
for i in range(N - 4, N):
    m = steps[i]
    e0, h0 = m[0][1], m[0][2]
    print('--- step %d%s' % (i, '' if prev0 is None else '   (%.2f ms after the previous step_start)' % prev0.elapsed_time(e0)))
    for name, e, h in m[1:]:
        print('   %-22s gpu %7.2f ms   host %6.2f ms' % (name, e0.elapsed_time(e), (h - h0) * 1e3))
    prev0 = e0
