import sys, time, torch
sys.path.insert(0, '/root/repo')
from oracle.build import build_model
from oracle.encoders import resnet152, roberta_large
import tell_amd
from tell_amd.data import synthetic_batch
torch.manual_seed(0)
model = build_model('faces_objects', resnet152(), roberta_large(), n_bert_layers=25).train()
for n, p in model.named_parameters():
    if n.startswith('resnet') or n.startswith('roberta'):
        p.requires_grad_(False)
batch = synthetic_batch(B=4, article_len=512, caption_len=33, faces_objects=True, seed=1234)
def clone(b):
    return {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()) for k, v in b.items()}
def step():
    model.zero_grad()
    model(**clone(batch))['loss'].backward()
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    step()
    t0 = time.time(); step(); print('threads', th, 'full fwd+bwd B=4: %.2f s' % (time.time() - t0), flush=True)
