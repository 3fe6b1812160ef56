"""Decoder-only training step (encoder outputs precomputed) - for profiling the small-kernel chain."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import hip, ops
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
tr = Trainer(model, device='cuda', async_update=False)
b = synthetic_batch(16, 512, 33, False, seed=1234, device='cuda')
model.train()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with hip.bound_stream():
    cap = {k: v.clone() for k, v in b['caption'].items()}
    _, tgt, ctx = model._forward(dict(b['context']), b['image'], cap)
    ctx = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ctx.items()}
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(steps + 2):
        if it == 2:
            e0.record()
        dec = model.decoder(cap, dict(ctx))
        ls, n = model.criterion(model.decoder.adaptive_softmax, dec, tgt)
        loss = (ls / math.log(2) / n.to(torch.float32)).reshape(())
        loss.backward()
        ops.join_wgrad_stream()
        tr._update()
    e1.record()
    torch.cuda.synchronize()
print('decoder fwd+loss+bwd+opt: %.3f ms/step' % (e0.elapsed_time(e1) / steps))
