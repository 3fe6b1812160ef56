"""3000 pipelined steps on two alternating synthetic batches: the loss must fall smoothly to ~0 (memorisation) and
stay finite - a soak test for rare races in the kernels / graph replays / stream schedule."""
import sys, torch, math
sys.path.insert(0, '.')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
kind = sys.argv[1] if len(sys.argv) > 1 else 'flattened'
fo = kind == 'faces_objects'
model = build_model(kind, weigh_bert=fo)
tr = Trainer(model, device='cuda', capture_after=1)
bs = [synthetic_batch(32 if fo else 16, 512, 33, fo, seed=1234 + 97 * i, device='cuda') for i in range(2)]
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
for i in range(3001):
    l = tr.train_one_batch(fresh(bs[i % 2]), next_batch=fresh(bs[(i + 1) % 2]))
    if i % 250 == 0:
        v = float(l)
        print(i, '%.6f' % v, 'finite' if math.isfinite(v) else 'NOT FINITE', flush=True)
w = tr.flat.flat
print('weights finite:', bool(torch.isfinite(w).all()), 'max |w|', float(w.abs().max()))
