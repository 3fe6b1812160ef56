"""Which steps of bench.py's variable-length loader leg replay a step graph, and what each batch shape costs: wraps
Trainer.train_one_batch with a synchronising timer (so the numbers are per-step GPU + host time, not pipelined)."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import tell_amd
from tell_amd.training import Trainer
orig = Trainer.train_one_batch
log = []


def timed(self, batch, next_batch=None):
    sg = self.step_graph
    before = sg.replays if sg is not None else 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(self, batch, next_batch=next_batch)
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
    sg = self.step_graph
    after = sg.replays if sg is not None else 0
    b = self._bucketed(batch)
    sig = (tuple(b['context']['roberta'].shape), tuple(b['caption']['roberta'].shape), tuple(b['face_embeds'].shape[:2]), tuple(b['obj_embeds'].shape[:2]))
    log.append((sig, after - before, dt))
    return out


Trainer.train_one_batch = timed
args = types.SimpleNamespace(batch=32)
r = bench.loader_bench(args, torch.device('cuda:0'), n_batches=22, warm=2, variable=True)
for i, (sig, rep, dt) in enumerate(log):
    print('%3d  %-60s replay %d  %6.1f ms' % (i, sig, rep, dt))
print({k: r[k] for k in ('value', 'ms_per_step', 'step_graph_replays', 'steps_total')})
