import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
tell_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = build_model('faces_objects').cuda().eval()
batch = synthetic_batch(32, 512, 33, True, device='cuda')
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}
beam = int(sys.argv[1]) if len(sys.argv) > 1 else 4
with tell_amd.hip.bound_stream():
    for _ in range(2):
        model.generate(**fresh(batch), beam_size=beam)
    torch.cuda.synchronize()
