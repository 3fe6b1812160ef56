import sys, collections, traceback, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
trainer = Trainer(model, device='cuda')
batch = synthetic_batch(16, 512, 33, False, device='cuda')
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
for _ in range(2):
    trainer.train_one_batch(fresh(batch))
torch.cuda.synchronize()
cnt = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        st = [s for s in traceback.extract_stack()[:-1] if 'transform-and-tell_amd' in s.filename]
        key = (name, tuple('%s:%d' % (s.filename.split('transform-and-tell_amd/')[-1], s.lineno) for s in st[-2:]))
        if name != 'contiguous' or not self.is_contiguous():
            cnt[key] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ['copy_', 'clone', 'contiguous', 'to', '__setitem__', 'float', 'fill_', 'zero_']:
    wrap(n)
for fn in ['zeros', 'zeros_like', 'cat', 'full']:
    orig = getattr(torch, fn)
    def mk(orig, fn):
        def f(*a, **k):
            st = [s for s in traceback.extract_stack()[:-1] if 'transform-and-tell_amd' in s.filename]
            cnt[(fn, tuple('%s:%d' % (s.filename.split('transform-and-tell_amd/')[-1], s.lineno) for s in st[-2:]))] += 1
            return orig(*a, **k)
        return f
    setattr(torch, fn, mk(orig, fn))
trainer.train_one_batch(fresh(batch))
torch.cuda.synchronize()
for (name, frames), n in cnt.most_common(30):
    print(n, name, ' <- '.join(frames))
