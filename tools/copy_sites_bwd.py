"""Where do the device copies of one training step come from (forward AND the autograd thread)?  Counts the calls of
Tensor.contiguous() that really copy, clone(), copy_() and to() by Python call site."""
import collections, sys, torch, traceback
sys.path.insert(0, '.')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
tr = Trainer(model, device='cuda')
b = synthetic_batch(16, 512, 33, False, seed=1234, device='cuda')
fresh = lambda: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
for _ in range(2):
    tr.train_one_batch(fresh())
sites = collections.Counter()


def site():
    st = traceback.extract_stack()
    f = next((f for f in reversed(st[:-2]) if 'tell' in f.filename and 'copy_sites' not in f.filename), st[0])
    return '%s:%d' % (f.filename.split('/')[-1], f.lineno)


orig = {n: getattr(torch.Tensor, n) for n in ('contiguous', 'clone', 'copy_', 'to')}


def wrap(name):
    f = orig[name]

    def g(self, *a, **k):
        if self.is_cuda and not (name == 'contiguous' and self.is_contiguous()):
            sites[(name, site(), tuple(self.shape))] += 1
        return f(self, *a, **k)
    return g


for n in orig:
    setattr(torch.Tensor, n, wrap(n))
tr.train_one_batch(fresh())
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(torch.Tensor, n, f)
for (name, s, shape), n in sorted(sites.items(), key=lambda kv: -kv[1])[:40]:
    print('%4d  %-12s %-28s %s' % (n, name, s, shape))
print('total:', sum(sites.values()))
