"""Two REAL data-parallel ranks on one GPU (gloo moves the gradients through the host): the multi-rank code path of the
trainer - token-count weighting, per-layer gradient exchange started during backward, final exchange, optimizer - with
different batches per rank.  Checks that both ranks hold bit-identical weights after every step (a gradient slice
exchanged before its last contribution would make them diverge) and prints a checksum to compare the bucketed schedule
with the single exchange (TELL_DP_BUCKETED=0).
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29650 tools/dp_two_ranks.py"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault('TELL_ALLREDUCE_FP32', '1')          # gloo: keep the wire dtype fp32
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)                                     # both ranks share the one GPU of the box
dist.init_process_group('gloo')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(77 + rank)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
for m in model.modules():                                    # no dropout: the two schedules must agree exactly
    for a in ('dropout', 'input_dropout', 'relu_dropout', 'weight_dropout', 'attention_dropout'):
        if isinstance(getattr(m, a, None), float):
            setattr(m, a, 0.0)
tr = Trainer(model, device='cuda', capture_after=1)
assert tr.dp and tr.world == 2
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
ok = True
for step in range(4):
    b = synthetic_batch(4, 128, 17 + 3 * rank, False, seed=500 + 10 * step + rank, device='cuda')   # ragged across ranks
    loss = tr.train_one_batch(fresh(b))
    tr.finish_update()
    torch.cuda.synchronize()
    flat = tr.flat.flat
    other = flat.clone()
    dist.broadcast(other, src=0)
    same = bool(torch.equal(other, flat))
    ok &= same
    chk = float(flat.double().abs().sum())
    if rank == 0:
        print('step %d  loss %.6f  weights identical on both ranks: %s  checksum %.10e  bucketed reduces so far: %d'
              % (step, float(loss), same, chk, tr.bucketed_reduces), flush=True)
flag = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print('RESULT', 'ranks consistent' if flag.item() == 1.0 else 'RANKS DIVERGED', flush=True)
dist.destroy_process_group()
