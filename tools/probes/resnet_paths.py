import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tell_amd
from tell_amd.models import resnet as R
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm()).item()
torch.manual_seed(0)
base = R.resnet152()
g = torch.Generator().manual_seed(1)
for m in base.modules():
    if isinstance(m, R._BN):
        m.weight.data.uniform_(0.5, 1.5, generator=g); m.bias.data.normal_(0, 0.2, generator=g)
sd = base.state_dict()
for B in (2, 8, 32):
    img = torch.randn(B, 3, 224, 224, device='cuda')
    outs = {}
    for name, dtype, implicit in (('fp32', torch.float32, False), ('bf16 im2col', torch.bfloat16, False), ('bf16 implicit', torch.bfloat16, True)):
        tell_amd.set_compute_dtype(dtype)
        m = R.resnet152(); m.load_state_dict(sd); m.cuda().train()
        ok = R.implicit_ok
        if not implicit: R.implicit_ok = lambda c, d: False
        try: outs[name] = m(img).float()
        finally: R.implicit_ok = ok
    print('B=%d train: bf16 im2col vs fp32 %.3e | bf16 implicit vs fp32 %.3e | implicit vs im2col %.3e' % (
        B, rel(outs['bf16 im2col'], outs['fp32']), rel(outs['bf16 implicit'], outs['fp32']), rel(outs['bf16 implicit'], outs['bf16 im2col'])), flush=True)
