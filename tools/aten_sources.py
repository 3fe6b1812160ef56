"""Which Python lines of the decoder half still launch ATen kernels (fills, adds, copies)?  One eager step of
tools/decoder_profile.py's setup under torch.profiler with stacks; prints device-kernel-launching aten ops grouped by the
innermost tell_amd / tests frame.  usage: TELL_STEP_GRAPH=0 python tools/aten_sources.py [model] [B]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TELL_STEP_GRAPH'] = '0'
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.models.transformer import EncodedBatch
from tell_amd.training import Trainer
kind = sys.argv[1] if len(sys.argv) > 1 else 'faces_objects'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tell_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
fo = kind == 'faces_objects'


class NoEncoder(torch.nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError('stub encoder')
    extract_features = forward


model = build_model(kind, NoEncoder(), NoEncoder(), weigh_bert=fo)
tr = Trainer(model, device='cuda', capture_after=1)
b = synthetic_batch(B, 512, 33, fo, seed=1234, device='cuda')
enc = EncodedBatch()
g = torch.Generator(device='cuda').manual_seed(5)
enc.stack = torch.randn(25, B, 512, 1024, device='cuda', generator=g).to(torch.bfloat16)
enc.x_image = torch.randn(B, 49, 2048, device='cuda', generator=g).abs().to(torch.bfloat16)
enc.article_mask = b['context']['roberta'] == 1
enc.static = True
fresh = lambda x: {k: (dict(v) if isinstance(v, dict) else v) for k, v in x.items()}
for _ in range(2):
    tr._prefetched = (b['image'], enc)
    tr.train_one_batch(fresh(b))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
tr._prefetched = (b['image'], enc)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_one_batch(fresh(b))
    torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=12):
    dev = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
    if not ev.key.startswith('aten::') or dev <= 0:
        continue
    frames = [f for f in ev.stack if 'tell_amd' in f or 'transform-and-tell' in f]
    frame = frames[0] if frames else (ev.stack[0] if ev.stack else '?')
    rows[(ev.key, frame.strip()[-120:])] += ev.count
for (name, frame), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print('%3d  %-24s %s' % (n, name, frame))
