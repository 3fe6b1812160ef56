"""world_size-2 data-parallel path on CPU (gloo): the device-independent DP pieces
(tell_amd/training/dp.py) must reproduce ONE process on the concatenated batch."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make():
    from oracle.build import build_decoder
    torch.manual_seed(0)
    return build_decoder('flattened', vocab_size=600, dim=64, heads=4, ffn=128, cutoff=(100, 300),
                         article_dim=64).eval()


def _batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, 600, (B, 7), generator=g)
    ids[:, 0] = 0
    tgt = torch.randint(4, 600, (B, 7), generator=g)
    n_pad = torch.randint(0, 4, (B,), generator=g)
    for b in range(B):
        if n_pad[b]:
            tgt[b, -int(n_pad[b]):] = 1                   # ragged token counts per rank
    ctx = {'image': torch.randn(5, B, 2048, generator=g), 'image_mask': torch.zeros(B, 5, dtype=torch.bool),
           'article': torch.randn(9, B, 64, generator=g), 'article_mask': torch.zeros(B, 9, dtype=torch.bool)}
    return ids, tgt, ctx


def _loss(dec, ids, tgt, ctx):
    from oracle.modules import AdaptiveLoss
    out = dec({'roberta': ids}, ctx)
    s, n = AdaptiveLoss(1)(dec.adaptive_softmax, out, tgt)
    return s / n, n


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import tell_amd  # noqa: F401
    from tell_amd.training import dp
    dec = _make()
    ids, tgt, ctx = _batch(3, 100 + rank)
    loss, n = _loss(dec, ids, tgt, ctx)
    w = dp.loss_weight(torch.tensor([float(n)]), dist, world)
    (loss * w.reshape(())).backward()
    params = [p for p in dec.parameters() if p.grad is not None]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dp.all_reduce_flat(flat, dist, bucket_elems=50000)       # several buckets
    flat /= world                                            # optimizer's grad_scale = 1/world
    if rank == 0:
        ret['flat'] = flat
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process_on_concatenated_batch():
    sys.path.insert(0, ROOT)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    dec = _make()
    parts = [_batch(3, 100), _batch(3, 101)]
    ids = torch.cat([p[0] for p in parts])
    tgt = torch.cat([p[1] for p in parts])
    ctx = {k: torch.cat([p[2][k] for p in parts], dim=0 if k.endswith('_mask') else 1) for k in parts[0][2]}
    loss, n = _loss(dec, ids, tgt, ctx)
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in dec.parameters() if p.grad is not None])
    torch.testing.assert_close(ret['flat'], ref, rtol=1e-4, atol=1e-6)
