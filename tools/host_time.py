import sys, time, torch
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
for _ in range(3):
    trainer.train_one_batch(fresh(batch))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    trainer.train_one_batch(fresh(batch))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) / 6 * 1e3, (t2 - t0) / 6 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
trainer.train_one_batch(fresh(batch))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
