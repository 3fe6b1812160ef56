"""Training steps with changing article / caption lengths and batch sizes: every new (B, S) signature of the
encoders is first run eagerly, then replayed; at most graphs.MAX_SIGNATURES are kept."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
tr = Trainer(model, device='cuda')
shapes = [(16, 512, 33), (16, 384, 25), (16, 512, 33), (8, 256, 17), (16, 384, 25), (16, 448, 29), (16, 320, 21), (12, 200, 12),
          (16, 512, 33), (16, 384, 25)]
batches = [synthetic_batch(B, S, T, False, seed=10 + i, device='cuda', variable=(i % 2 == 1)) for i, (B, S, T) in enumerate(shapes)]
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
t0 = time.perf_counter()
for i, b in enumerate(batches):
    nb = batches[i + 1] if i + 1 < len(batches) else None
    loss = tr.train_one_batch(fresh(b), next_batch=nb)
    print(i, shapes[i], 'loss %.4f' % float(loss), flush=True)
torch.cuda.synchronize()
g = model.__dict__['_roberta_graph']
print('roberta graph states:', [(k[0], v['state']) for k, v in g.entries.items()])
print('ok, %.1f s' % (time.perf_counter() - t0))
