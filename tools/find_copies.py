import sys, collections, torch
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_one_batch(fresh(batch))
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::cat'):
        frames = [s for s in e.stack if 'transform-and-tell_amd' in s or 'bench' in s][:2]
        cnt[(e.name, tuple(frames))] += 1
for (name, frames), n in cnt.most_common(25):
    print(n, name, ' <- '.join(f.split('transform-and-tell_amd/')[-1] for f in frames))
