"""GPU time of each frozen encoder alone (events, train mode, bf16, B=16)."""
import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import hip
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False).to('cuda').train()
b = synthetic_batch(16, 512, 33, False, seed=1234, device='cuda')
ids, img = b['context']['roberta'], b['image']
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad(), hip.bound_stream():
    print('roberta-large fwd (16x512):  %.3f ms' % t(lambda: model.roberta.extract_features(ids, return_all_hiddens=True)))
    print('resnet-152 fwd (16x224x224): %.3f ms (graph replay)' % t(lambda: model._run_resnet(img)))
    print('resnet-152 fwd eager:        %.3f ms' % t(lambda: model.resnet(img)))
