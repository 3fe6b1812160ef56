"""Un-profiled GPU timeline of the pipelined configs[2] training step (rocprofv3 serialises the queues; this does not):
HIP timing events around each stream's share of a step - the RoBERTa graph replay and the ResNet graph replay of batch
N + 1 on their streams, the decoder step graph of batch N on the training stream - over steady-state steps, and the same
three pieces ALONE on an idle GPU.  Prints, per configuration: the step period, when each piece starts / ends inside it, how
much longer it takes there than alone (its stretch = the CU-time it waited for), and the schedule's packing = (sum of the
alone times) / period (1.0 = the chip is never shared, i.e. a serial schedule; the pieces need 256 CUs each for most of
their launches, so this is the number that says how much of the chip's time the overlap recovers).
  python tools/step_timeline.py [steps]            (TELL_RESNET_STREAM=main, TELL_Q4_DYNAMIC=1 select the variants)"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.hip.require_gpu()
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
model = build_model('faces_objects', weigh_bert=True)
tr = Trainer(model, device='cuda', capture_after=1)
batches = [synthetic_batch(32, 512, 33, True, seed=1234 + 97 * i, device='cuda') for i in range(2)]
marks = {}


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    return e


def wrap(obj, attr, name):
    f = getattr(obj, attr)

    def g(*a, **k):
        e0 = ev()
        r = f(*a, **k)
        marks.setdefault(name, []).append((e0, ev()))
        return r
    setattr(obj, attr, g)


wrap(model, '_run_roberta', 'roberta(N+1)')
wrap(model, '_run_resnet', 'resnet(N+1)')
if tr.step_graph is not None:
    wrap(tr.step_graph, 'run', 'decoder(N)')
else:
    wrap(tr, '_eager_step', 'decoder(N)')


def fresh(b):
    return {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}


for i in range(8):                                       # warm-up: captures
    tr.train_one_batch(fresh(batches[i % 2]), next_batch=batches[(i + 1) % 2])
for i in range(60):                                      # clock settling
    tr.train_one_batch(fresh(batches[i % 2]), next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()
marks.clear()
starts = []
for i in range(N):
    starts.append(ev())
    tr.train_one_batch(fresh(batches[i % 2]), next_batch=batches[(i + 1) % 2])
end = ev()
torch.cuda.synchronize()
period = starts[0].elapsed_time(end) / N
med = statistics.median
rows = {}
for name, lst in marks.items():
    lst = lst[-N:]
    rows[name] = (med([starts[i].elapsed_time(a) for i, (a, b) in enumerate(lst)]), med([a.elapsed_time(b) for a, b in lst]))
# ---- the three pieces alone on an idle GPU
alone = {}
enc = [model.encode(b['context'], b['image']) for b in batches]
torch.cuda.synchronize()
for name in ('roberta(N+1)', 'resnet(N+1)'):
    marks.pop(name, None)
for i in range(6):
    model._run_roberta(batches[i % 2]['context']['roberta'], i)
    torch.cuda.synchronize()
for i in range(6):
    model._run_resnet(batches[i % 2]['image'], i)
    torch.cuda.synchronize()
alone['roberta(N+1)'] = med([a.elapsed_time(b) for a, b in marks['roberta(N+1)'][-4:]])
alone['resnet(N+1)'] = med([a.elapsed_time(b) for a, b in marks['resnet(N+1)'][-4:]])
marks.pop('decoder(N)', None)
for i in range(6):
    tr._prefetched = (batches[i % 2]['image'], enc[i % 2])
    tr.train_one_batch(fresh(batches[i % 2]))
    torch.cuda.synchronize()
alone['decoder(N)'] = med([a.elapsed_time(b) for a, b in marks['decoder(N)'][-4:]])
cfg = 'resnet stream = %s, q4 tile queue = %s' % (os.environ.get('TELL_RESNET_STREAM', 'own'),
                                                 'per-XCD counters' if tell_amd.hip.get_option('q4_dynamic') else 'static lists')
print('# configs[2] step timeline (%s); %d steady-state steps, medians, ms' % (cfg, N))
print('step period %.3f ms = %.1f samples/s' % (period, 32e3 / period))
print('%-14s %8s %8s %8s %8s %8s' % ('piece', 'start', 'end', 'in-step', 'alone', 'stretch'))
tot = 0.0
for name in ('roberta(N+1)', 'resnet(N+1)', 'decoder(N)'):
    st, du = rows[name]
    tot += alone[name]
    print('%-14s %8.2f %8.2f %8.2f %8.2f %8.2f' % (name, st, st + du, du, alone[name], du - alone[name]))
print('sum of the alone times %.2f ms; critical piece alone %.2f ms; packing = sum / period = %.3f (1.0 = serial, %.3f = perfect overlap)'
      % (tot, max(alone.values()), tot / period, tot / max(alone.values())))
