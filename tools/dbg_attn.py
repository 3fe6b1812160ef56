import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import ops
DEV = 'cuda'
def run(T, B, S, use_mask, dtype, has_extra, mode='rand'):
    E, H = 1024, 16
    g = torch.Generator().manual_seed(1)
    q = torch.randn(T, B, E, generator=g) * 0.5
    k = torch.randn(S, B, E, generator=g)
    v = torch.randn(S, B, E, generator=g)
    mask = (torch.rand(B, S, generator=g) < 0.3) if use_mask else None
    if use_mask and mode == 'zeros': mask = torch.zeros(B, S, dtype=torch.bool)
    if use_mask and mode == 'first': mask = torch.zeros(B, S, dtype=torch.bool); mask[:, 0] = True
    if use_mask and mode == 'tail': mask = torch.zeros(B, S, dtype=torch.bool); mask[:, S//2:] = True
    if use_mask and mode == 'k5': mask = torch.zeros(B, S, dtype=torch.bool); mask[:, 5] = True
    if use_mask and mode == 'k40': mask = torch.zeros(B, S, dtype=torch.bool); mask[:, 40] = True
    qd, kd, vd = [t.to(DEV, dtype) for t in (q, k, v)]
    md = mask.to(DEV).to(torch.uint8) if use_mask else None
    bk = torch.randn(1, 1, E, generator=g).to(DEV) if has_extra else None
    bv = torch.randn(1, 1, E, generator=g).to(DEV) if has_extra else None
    out = ops.attention(qd, kd, vd, md, bk, bv, H, has_zero=has_extra)
    hd = E // H
    kk, vv = kd.float().cpu(), vd.float().cpu()
    if has_extra:
        kk = torch.cat([kk, bk.to(dtype).float().cpu().expand(1, B, E), torch.zeros(1, B, E)])
        vv = torch.cat([vv, bv.to(dtype).float().cpu().expand(1, B, E), torch.zeros(1, B, E)])
    S1 = kk.shape[0]
    qc = qd.float().cpu().reshape(T, B * H, hd).transpose(0, 1)
    kc = kk.reshape(S1, B * H, hd).transpose(0, 1)
    vc = vv.reshape(S1, B * H, hd).transpose(0, 1)
    sc = torch.bmm(qc, kc.transpose(1, 2))
    if use_mask:
        full = torch.cat([mask, torch.zeros(B, S1 - S, dtype=torch.bool)], 1)
        sc = sc.view(B, H, T, S1).masked_fill(full[:, None, None, :], float('-inf')).view(B * H, T, S1)
    ref = torch.bmm(torch.softmax(sc, -1), vc).transpose(0, 1).reshape(T, B, E)
    o = out.float().cpu()
    errs = [((o[:, b] - ref[:, b]).norm() / ref[:, b].norm()).item() for b in range(B)]
    errt = [((o[t0:t0+32] - ref[t0:t0+32]).norm() / ref[t0:t0+32].norm()).item() for t0 in range(0, T, 32)]
    print(mode, 'T%d B%d S%d mask=%s %s extra=%s: per-b %s per-qblock %s' % (T, B, S, use_mask, dtype, has_extra,
          ['%.3g' % e for e in errs], ['%.3g' % e for e in errt]))
for dtype in (torch.float32, torch.bfloat16):
    run(128, 1, 128, True, dtype, False, 'zeros')
    run(128, 1, 128, True, dtype, False, 'rand')
    run(160, 2, 200, True, dtype, True, 'rand')
