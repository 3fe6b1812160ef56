"""Marginal cost of each stream of the pipelined step: the same steps with the ResNet and / or RoBERTa replay replaced
by its cached output (no kernels), and with the decoder alone."""
import sys, time, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
from tell_amd.build import build_model
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
kind = sys.argv[1] if len(sys.argv) > 1 else 'faces_objects'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
fo = kind == 'faces_objects'
model = build_model(kind, weigh_bert=fo)
tr = Trainer(model, device='cuda', capture_after=1)
batches = [synthetic_batch(B, 512, 33, fo, seed=1234 + i, device='cuda') for i in range(2)]
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
run_res, run_rob = model._run_resnet, model._run_roberta
cache = {}


def cached(name, f):
    def g(x):
        if name not in cache:
            cache[name] = f(x)
        return cache[name]
    return g


def measure(label, steps=30):
    for i in range(6):
        tr.train_one_batch(fresh(batches[i % 2]), next_batch=fresh(batches[(i + 1) % 2]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_one_batch(fresh(batches[i % 2]), next_batch=fresh(batches[(i + 1) % 2]))
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print('%-34s %6.2f ms/step   (host issue %5.2f ms)' % (label, dt, issued / steps * 1e3), flush=True)


measure('full')
model._run_resnet = cached('res', run_res)
measure('no ResNet kernels')
model._run_resnet = run_res
model._run_roberta = cached('rob', run_rob)
measure('no RoBERTa kernels')
model._run_resnet = cached('res', run_res)
measure('decoder + optimizer only')
