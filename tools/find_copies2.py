import sys, collections, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd.models.resnet import resnet152
tell_amd.set_compute_dtype(torch.bfloat16)
m = resnet152().cuda().train()
for p in m.parameters(): p.requires_grad_(False)
x = torch.randn(4, 3, 224, 224, device='cuda')
m(x); m(x); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    m(x); torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if 'emcpy' in e.name or 'emset' in e.name or 'copyBuffer' in e.name or e.name.startswith('hipMem'):
        c[e.name] += 1
print(c.most_common(10))
c2 = collections.Counter()
for e in prof.events():
    if e.name.startswith('aten::'):
        c2[(e.name, str(e.input_shapes)[:60])] += 1
print(c2.most_common(12))
