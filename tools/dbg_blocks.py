import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import tell_amd, seeded
from tell_amd.build import build_decoder
from tell_amd.modules import AdaptiveLoss
from tell_amd.models import decoders
DEC_KW = dict(vocab_size=600, dim=64, heads=4, ffn=128, cutoff=(100, 300))
kind = sys.argv[1] if len(sys.argv) > 1 else 'flattened'
fx = seeded.load_npz('tests/golden/decoder_%s.npz' % kind)
tell_amd.set_compute_dtype(torch.bfloat16)
res = {}
for blocks_on in (False, True):
    decoders._BLOCKS = blocks_on
    dec = build_decoder(kind, article_dim=64 if kind.startswith('flattened') else 1024, **DEC_KW)
    dec.load_state_dict(fx['sd'], strict=False)
    dec.cuda().train()
    for m in dec.modules():
        for attr in ('dropout', 'input_dropout', 'relu_dropout', 'weight_dropout'):
            if hasattr(m, attr) and isinstance(getattr(m, attr), float):
                setattr(m, attr, 0.0)
    ins = fx['in']
    ctx = {}
    for k, v in ins.items():
        if k in ('ids', 'target'):
            continue
        ctx[k] = v.cuda() if v.dtype == torch.bool else v.cuda().to(torch.bfloat16)
    out = dec({'roberta': ins['ids'].cuda()}, ctx)
    loss, n = AdaptiveLoss(1)(dec.adaptive_softmax, out, ins['target'].cuda())
    (loss / n.float()).sum().backward()
    res[blocks_on] = {k: p.grad.float().cpu().clone() for k, p in dec.named_parameters() if p.grad is not None}
    print('blocks', blocks_on, 'loss', float(loss))
for k in res[False]:
    a, b = res[False][k], res[True][k]
    r = float((a - b).norm() / (a.norm() + 1e-30))
    if r > 0.05:
        print('%-70s rel %.3f  shape %s  tail zero rows: %d' % (k, r, tuple(a.shape), int((b.reshape(b.shape[0], -1).abs().sum(1) == 0).sum())))
print('done')
