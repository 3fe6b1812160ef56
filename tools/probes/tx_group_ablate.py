"""The backward pass's weight gradients as grouped launches, alone: the four article K|V ones (16384-row reduction) and
twenty 1024^3 ones.  (With the operand loads compiled out of the kernel body the first group ran at 1152 TFLOP/s instead
of 605: it is bound by operand re-reads through L2 - 4 GB per step with 128x128 tiles - not by staging or barriers.)"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
import tell_amd
from tell_amd import ops
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 10
# the article K|V weight gradients of the four layers: dW[2048,1024] += dKV^T[2048 x 16384] X[16384 x 1024], + 20 small ones
K = 16384
big = [(torch.randn(K, 2048, device='cuda').bfloat16(), torch.randn(K, 1024, device='cuda').bfloat16(), torch.zeros(2048, 1024, device='cuda')) for _ in range(4)]
small = [(torch.randn(1024, 1024, device='cuda').bfloat16(), torch.randn(1024, 1024, device='cuda').bfloat16(), torch.zeros(1024, 1024, device='cuda')) for _ in range(20)]
for name, probs in (('4 x article kv wgrad (K=16384)', big), ('20 x 1024^3 wgrad', small)):
    fl = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _ in probs)
    t = timed(lambda: ops.gemm_grouped([dict(a=a, b=b, out=o, form='tn', accumulate=True) for a, b, o in probs]))
    print('%-34s %7.1f us  %5.0f TFLOP/s' % (name, t, fl / t * 1e-6))
