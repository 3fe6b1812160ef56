"""Two REAL data-parallel ranks of the HIP Trainer (both on cuda:0, gloo carries the collectives) against ONE process on the
concatenated batch ("N-GPU result == 1-GPU result on the concatenated batch").  fp32, stand-in encoders
(no cross-sample coupling), dropout off, ragged captions so that the ranks' token counts differ.  Rank 0 also trains a
second copy of the model single-process on the concatenation of both ranks' batches; after every step the data-parallel
weights must equal the single-process weights.  Runs the eager schedule (bucketed exchange during backward, loss
weighted before backward) or the step-graph schedule (TELL_STEP_GRAPH, gradient weighted on the way to the wire).
DP_EQ_WIRE=bf16 + DP_EQ_TOL bound the drift of the bf16 wire format.
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29660 tools/dp_equivalence.py"""
import copy, os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
from test_gpu_train import _Res, _Rob, _no_dropout, KW
tell_amd.set_compute_dtype(torch.float32)
torch.manual_seed(0)
model = build_model('faces_objects', _Res(True), _Rob(1024), n_bert_layers=3, **KW)
_no_dropout(model)
for m in model.modules():
    if isinstance(getattr(m, 'dropout', None), float):
        m.dropout = 0.0
single = copy.deepcopy(model) if rank == 0 else None
ocfg = dict(lr=5e-3, warmup=0.5, t_total=8, b1=0.9, b2=0.98, e=1e-6, weight_decay=1e-5, max_grad_norm=0.1)
# DP_EQ_WIRE=bf16: the production wire format (gradients rounded to bf16 for the exchange, BertAdam reads the reduced
# bf16 buffer) while everything else stays fp32 - the drift against the single process is then the wire's rounding alone
wire = torch.bfloat16 if os.environ.get('DP_EQ_WIRE') == 'bf16' else None
tr = Trainer(model, dict(ocfg), device='cuda', allreduce_dtype=wire)
assert tr.dp and tr.world == 2
ts = Trainer(single, dict(ocfg), device='cuda', data_parallel=False) if rank == 0 else None
STEPS = int(os.environ.get('DP_EQ_STEPS', '5'))


def dev(b):
    return {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in b.items()}


def cat(a, b):
    out = {}
    for k in a:
        if isinstance(a[k], dict):
            out[k] = {kk: torch.cat([a[k][kk], b[k][kk]]) for kk in a[k]}
        else:
            out[k] = torch.cat([a[k], b[k]])
    return out


worst = 0.0
for step in range(STEPS):
    parts = [synthetic_batch(3, 24, 12, True, seed=900 + 10 * step + r, vocab=600, cutoffs=(100, 300), variable=True)
             for r in range(world)]
    loss = tr.train_one_batch(dev(parts[rank]))
    tr.finish_update()
    torch.cuda.synchronize()
    if rank == 0:
        ls = ts.train_one_batch(dev(cat(parts[0], parts[1])))
        torch.cuda.synchronize()
        num = float((tr.flat.flat - ts.flat.flat).norm())
        den = float(ts.flat.flat.norm())
        worst = max(worst, num / den)
        print('step %d  dp loss(rank0) %.6f  single loss %.6f  |w_dp - w_single| / |w| = %.3e  graph replays %d'
              % (step, float(loss), float(ls), num / den, tr.step_graph.replays if tr.step_graph else 0), flush=True)
flag = torch.tensor([worst])
dist.broadcast(flag, src=0)
if rank == 0:
    print('RESULT', 'dp == single' if worst < float(os.environ.get('DP_EQ_TOL', '2e-5')) else 'DP DIFFERS', 'worst %.3e' % worst,
          flush=True)
dist.destroy_process_group()
