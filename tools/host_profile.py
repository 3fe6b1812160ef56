"""cProfile of the host side of pipelined training steps (where do the ~13 ms of Python per step go?)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
from tell_amd.training import Trainer
tell_amd.set_compute_dtype(torch.bfloat16)
tell_amd.manual_seed(1234)
torch.manual_seed(0)
model = build_model('flattened', weigh_bert=False)
tr = Trainer(model, device='cuda')
bs = [synthetic_batch(16, 512, 33, False, seed=1234 + i, device='cuda') for i in range(2)]
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}
for i in range(4):
    tr.train_one_batch(fresh(bs[i % 2]), next_batch=bs[(i + 1) % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
N = 10
for i in range(4, 4 + N):
    tr.train_one_batch(fresh(bs[i % 2]), next_batch=bs[(i + 1) % 2])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
