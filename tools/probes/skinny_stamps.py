"""Where does a skinny linear's launch time go?  (probe build only: PROBES=1 csrc/build.sh, TELL_LIB=.../libtell_hip_probes.so)

Every skinny_mfma_kernel workgroup of the captured decode step stamps the 100 MHz wall clock at entry, after its K loop and at
exit (csrc/decode.hip under TELL_PROBES; options sk_stamp_ptr / sk_stamp_slots).  A captured launch keeps its slot, so the
last graph replay of a generation leaves one consistent step in the buffer.  Per launch this prints

  dur     first workgroup entry -> last workgroup exit
  gap     last exit of the PREVIOUS skinny launch -> first entry of this one (other kernels in between are named by count only)
  ramp    first entry -> last entry (how long the dispatcher takes to get every workgroup on a CU)
  life    mean workgroup lifetime, split into K loop and epilogue; the epilogue again into e:lds (partial tiles into LDS + the
          barrier), e:sum (the sums over the waves, row statistics) and e:out (residual arithmetic + stores + exit)
  p50/p90/max lifetimes, workgroups per XCD

usage: python tools/probes/skinny_stamps.py [--beam 4] [--batch 32]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SLOT = (1 + 2048) * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--beam', type=int, default=4)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--slots', type=int, default=4096)
    args = ap.parse_args()
    import tell_amd
    from tell_amd import hip
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    assert hip.lib().tell_probe_build(), 'needs the probe build: TELL_LIB=<csrc>/libtell_hip_probes.so'
    dev = torch.device('cuda:0')
    tell_amd.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = build_model('faces_objects').to(dev).eval()
    batch = synthetic_batch(args.batch, 512, 33, True, seed=4321, device=dev)
    stamps = torch.zeros(args.slots * SLOT, dtype=torch.int64, device=dev)
    hip.set_option('sk_stamp_slots', args.slots)
    hip.set_option('sk_stamp_ptr', stamps.data_ptr())
    with hip.bound_stream():
        model.generate(**batch, beam_size=args.beam)
        torch.cuda.synchronize()
        stamps.zero_()
        torch.cuda.synchronize()
        model.generate(**batch, beam_size=args.beam)          # replays only: the captured slots are refreshed
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.uint64).reshape(args.slots, 1 + 2048, 4)
    rows = []
    for i in range(args.slots):
        g = int(s[i, 0, 0])
        if g == 0:
            continue
        gx, gy, gz = g & 0xffff, (g >> 16) & 0xffff, (g >> 32) & 0xffff
        n = gx * gy * gz
        sh = int(s[i, 0, 1])
        M, N, K = sh & 0xfffff, (sh >> 20) & 0xfffff, sh >> 40
        t = int(s[i, 0, 2])
        code = 'RT%d act%d U%d%s%s' % (t & 255, (t >> 8) & 255, (t >> 16) & 255, ' fold' if (t >> 24) & 1 else '', ' split' if (t >> 25) & 1 else '')
        w = s[i, 1:1 + n].astype(np.int64)
        rows.append(dict(slot=i, grid=(gx, gy, gz), M=M, N=N, K=K, code=code, cn=int(s[i, 0, 3]), t0=w[:, 0], t1=w[:, 1], t2=w[:, 2],
                         xcc=(w[:, 3] >> 32) & 0xf, hw=w[:, 3] & 0xffff, d_sync=(w[:, 3] >> 36) & 0xfff, d_calc=(w[:, 3] >> 48) & 0xfff))
    rows.sort(key=lambda r: r['t0'].min())
    # the last replay: the launches whose stamps lie within one step of the newest one
    newest = max(r['t2'].max() for r in rows)
    rows = [r for r in rows if newest - r['t0'].min() < 100000]           # 1 ms in 10 ns ticks
    print('# %d skinny launches of the last replayed step (beam %d, %d rows); times in us (wall clock, 10 ns ticks)' % (len(rows), args.beam, args.batch * args.beam))
    print('%-22s %-16s %5s %5s %5s | %6s %6s %6s | %6s %6s %6s | %6s %6s %6s | %5s %5s %5s | %s' % (
        'kernel', 'grid', 'M', 'N', 'K', 'dur', 'gap', 'ramp', 'life', 'kloop', 'epi', 'p50', 'p90', 'max', 'e:lds', 'e:sum', 'e:out',
        'wg/xcd  distinct CUs'))
    prev_end = None
    tot = dict(dur=0.0, gap=0.0, ramp=0.0, life=0.0)
    for r in rows:
        t0, t1, t2 = r['t0'], r['t1'], r['t2']
        dur = (t2.max() - t0.min()) / 100.0
        gap = (t0.min() - prev_end) / 100.0 if prev_end is not None else float('nan')
        ramp = (t0.max() - t0.min()) / 100.0
        life = (t2 - t0) / 100.0
        kl = (t1 - t0).mean() / 100.0
        ep = (t2 - t1).mean() / 100.0
        per = np.bincount(r['xcc'].astype(np.int64), minlength=8)
        cu = len(set(zip(r['xcc'].tolist(), ((r['hw'] >> 8) & 0xf).tolist(), ((r['hw'] >> 13) & 0x7).tolist(), ((r['hw'] >> 12) & 1).tolist())))
        e_lds, e_sum = r['d_sync'].mean() / 100.0, r['d_calc'].mean() / 100.0     # epilogue: partial tiles to LDS + barrier | sums
        print('%-22s %-16s %5d %5d %5d | %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f | %5.2f %5.2f %5.2f | %s  %d' % (
            r['code'], 'x'.join(map(str, r['grid'])) + ' cn%d' % r['cn'], r['M'], r['N'], r['K'], dur, gap, ramp, life.mean(), kl, ep,
            np.percentile(life, 50), np.percentile(life, 90), life.max(), e_lds, e_sum, ep - e_lds - e_sum, '/'.join(map(str, per)), cu))
        prev_end = t2.max()
        tot['dur'] += dur
        tot['ramp'] += ramp
        tot['life'] += life.mean()
        if gap == gap:
            tot['gap'] += gap
    for r in rows[:6]:
        print('# XCC_ID of workgroups 0..23 of grid %s: %s' % ('x'.join(map(str, r['grid'])), ' '.join(map(str, r['xcc'][:24].tolist()))))
    print('# sums: dur %.1f us, gaps (incl. the other kernels in between) %.1f us, ramps %.1f us, mean lifetimes %.1f us' % (
        tot['dur'], tot['gap'], tot['ramp'], tot['life']))


if __name__ == '__main__':
    main()
