"""List the Python call sites whose ATen ops launch device copies / elementwise kernels in one training step."""
import collections, sys, torch, traceback
sys.path.insert(0, '.')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
from torch.utils._python_dispatch import TorchDispatchMode

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
SKIP = ('aten.view', 'aten.reshape', 'aten._unsafe_view', 'aten.detach', 'aten.transpose', 'aten.t.', 'aten.slice',
        'aten.select', 'aten.empty', 'aten.as_strided', 'aten.alias', 'aten.unsqueeze', 'aten.squeeze', 'aten.expand',
        'aten.permute', 'aten.new_empty', 'aten.empty_like', 'aten.split', 'aten.unbind', 'aten.is_', 'aten.sym_',
        'aten.stride', 'aten.size', 'aten._local_scalar', 'aten.lift_fresh', 'aten.narrow', 'aten.unfold')

class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            st = traceback.extract_stack()
            site = next((f for f in reversed(st) if 'tell' in f.filename and 'torch_copy_sites' not in f.filename), st[0])
            sites[(name, '%s:%d' % (site.filename.split('/')[-1], site.lineno))] += 1
        return func(*args, **(kwargs or {}))

with Mode():
    tr.train_one_batch(fresh())
torch.cuda.synchronize()
for (name, site), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print('%4d  %-40s %s' % (n, name, site))
print('total ATen ops with kernels:', sum(sites.values()))
