"""Decoder half of the optimisation step alone (decoder fwd + loss + bwd + BertAdam; the frozen encoders replaced by
static random outputs of the real shape), replayed through the trainer's step graph - the command behind
profiles/*_decoder_kernel_stats.txt:  rocprofv3 --kernel-trace --stats -- python tools/decoder_profile.py [model] [B] [steps]
Every kernel in the trace after the warm-up belongs to the decoder step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.models.transformer import EncodedBatch
from tell_amd.training import Trainer
kind = sys.argv[1] if len(sys.argv) > 1 else 'faces_objects'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
eager = os.environ.get('TELL_STEP_GRAPH') == '0'
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
fo = kind == 'faces_objects'


class NoEncoder(torch.nn.Module):          # never called: the features are handed to the trainer directly
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
ev = []
for it in range(steps + 3):
    tr._prefetched = (b['image'], enc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    loss = tr.train_one_batch(fresh(b))
    e1.record()
    ev.append((e0, e1))
torch.cuda.synchronize()
d = sorted(a.elapsed_time(c) for a, c in ev[3:])
print('decoder step (%s, B=%d, %s): median %.3f ms  min %.3f ms  loss %.4f  graph replays %d'
      % (kind, B, 'eager' if eager else 'hipGraph', d[len(d) // 2], d[0], float(loss), tr.step_graph.replays))
gf = {'faces_objects': 47.0, 'flattened': 37.9}[kind]
print('decoder MFMA fraction: %.4f (%.1f GF/sample x %d / %.3f ms / 2500 TFLOP/s)'
      % (gf * B / d[len(d) // 2] / 2500.0, gf, B, d[len(d) // 2]))
