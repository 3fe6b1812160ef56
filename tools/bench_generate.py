"""Caption generation throughput, full faces+objects model (BASELINE configs[4] shape): greedy (what the
reference does) and beam 4 (what configs[4] asks for).  Compares the K/V-cached static-batch generator with the reference's
control flow (per-step K/V recomputation + active-row compaction)."""
import sys, time, json, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd.build import build_model
from tell_amd.data import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tell_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = build_model('faces_objects').cuda().eval()
batch = synthetic_batch(B, 512, 33, True, device='cuda')
fresh = lambda b: {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}
res = {}
with tell_amd.hip.bound_stream():
    for name, fast, reps, beam in (('cached', True, 3, 1), ('beam4_cached', True, 2, 4), ('reference_flow', False, 1, 1)):
        model.fast_generation = fast
        out = model.generate(**fresh(batch), beam_size=beam)        # warm-up (weight casts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = model.generate(**fresh(batch), beam_size=beam)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[name] = {'captions_per_s': round(B / dt, 2), 's_per_batch': round(dt, 3), 'steps': out['gen_ids'].shape[1] - 1}
print(json.dumps({'batch': B, 'dtype': 'bf16', 'includes': 'ResNet-152 + RoBERTa-large encoders + decode', **res}))
