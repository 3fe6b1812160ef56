"""Per-phase GPU time of one training step (events on the main stream; encoders serialised by default)."""
import math, os, sys, time, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import hip, ops
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
from tell_amd.build import build_model
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
tr = Trainer(model, device='cuda')
batches = [synthetic_batch(16, 512, 33, False, seed=1234 + i, device='cuda') for i in range(2)]
def fresh(b):
    return {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
for i in range(3):
    tr.train_one_batch(fresh(batches[i % 2]))
torch.cuda.synchronize()
names = ['zero_grad', 'encoders', 'decoder_fwd', 'loss_fwd', 'backward', 'optimizer']
acc = dict.fromkeys(names, 0.0)
hacc = dict.fromkeys(names, 0.0)
N = 6
for it in range(N):
    b = fresh(batches[it % 2])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    ht = []
    torch.cuda.synchronize()
    with hip.bound_stream():
        model.train()
        ev[0].record(); ht.append(time.perf_counter()); tr.flat.zero_grad()
        ev[1].record(); ht.append(time.perf_counter()); cap_ids, tgt, ctx = model._forward(b['context'], b['image'], b['caption'])
        ev[2].record(); ht.append(time.perf_counter()); dec = model.decoder(b['caption'], ctx)
        ev[3].record(); ht.append(time.perf_counter())
        ls, n = model.criterion(model.decoder.adaptive_softmax, dec, tgt)
        loss = (ls / math.log(2) / n.to(torch.float32)).reshape(())
        ev[4].record(); ht.append(time.perf_counter()); loss.backward()
        ev[5].record(); ht.append(time.perf_counter()); tr.optimizer.step(grad_scale=1.0)
        ev[6].record(); ht.append(time.perf_counter())
    torch.cuda.synchronize()
    for i, nme in enumerate(names):
        acc[nme] += ev[i].elapsed_time(ev[i + 1]); hacc[nme] += (ht[i + 1] - ht[i]) * 1e3
tot = 0
for nme in names:
    print('%-12s gpu %7.3f ms   host-issue %7.3f ms' % (nme, acc[nme] / N, hacc[nme] / N)); tot += acc[nme] / N
print('%-12s %7.3f ms' % ('total', tot))
