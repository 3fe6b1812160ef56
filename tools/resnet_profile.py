"""ResNet-152 trunk alone (train-mode BatchNorm, bf16, hipGraph replay) - the command behind profiles/*_resnet_kernel_stats.txt:
rocprofv3 --kernel-trace --stats -- python tools/resnet_profile.py [B] [steps] [train|eval]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tell_amd
from tell_amd import graphs
from tell_amd.models.resnet import resnet152
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else 'train'
tell_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = resnet152().cuda()
net.train(mode == 'train')
g = graphs.GraphedCall(net, 'resnet152', buffers=1)
x = torch.randn(B, 3, 224, 224, device='cuda')
ev = []
with tell_amd.hip.bound_stream():
    for it in range(steps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = g(x)
        e1.record()
        ev.append((e0, e1))
torch.cuda.synchronize()
d = sorted(a.elapsed_time(b) for a, b in ev[3:])
ms = d[len(d) // 2]
print('ResNet-152 %s B=%d: median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (23 GF/img), out %s finite %s' % (
    mode, B, ms, d[0], 23.0 * B / ms, tuple(y.shape), bool(torch.isfinite(y.float()).all())))
