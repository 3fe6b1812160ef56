"""torch.profiler view of one pipelined training step: ATen ops (all threads) by count - finds hidden copies / fills."""
import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
from torch.profiler import profile, ProfilerActivity
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
tr = Trainer(model, device='cuda')
bs = [synthetic_batch(16, 512, 33, False, seed=1234 + i, device='cuda') for i in range(2)]
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
for i in range(3):
    tr.train_one_batch(fresh(bs[i % 2]), next_batch=fresh(bs[(i + 1) % 2]))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.train_one_batch(fresh(bs[1]), next_batch=fresh(bs[0]))
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
agg = {}
for r in rows:
    if r.key.startswith('aten::') and r.key.split('::')[1] in ('copy_', 'clone', 'contiguous', 'add', 'add_', 'zeros', 'fill_', 'zero_', 'to', '_to_copy', 'mul', 'sum', 'cat', 'empty_like', 'view_as'):
        agg[(r.key, str(r.input_shapes)[:90])] = agg.get((r.key, str(r.input_shapes)[:90]), 0) + r.count
for (k, sh), n in sorted(agg.items(), key=lambda kv: -kv[1])[:40]:
    print('%4d  %-16s %s' % (n, k, sh))

import collections
mem = collections.Counter()
for e in prof.events():
    n = e.name
    if 'emcpy' in n or 'emset' in n or 'copyBuffer' in n:
        mem[n] += 1
print('memcpy/memset-like events:', dict(mem))
# CPU-side callers of hipMemcpyAsync
callers = collections.Counter()
for e in prof.events():
    if e.name in ('hipMemcpyAsync', 'hipMemcpyWithStream', 'hipMemcpyDtoDAsync', 'hipMemcpyHtoDAsync'):
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        callers[(e.name, ' <- '.join(chain))] += 1
for (n, c), k in callers.most_common(25):
    print('%4d  %s  %s' % (k, n, c))
