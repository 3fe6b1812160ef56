import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import ops
shapes = [  # (M, N, K, out_f32, accumulate, label)
    (8192, 4096, 1024, False, False, 'roberta fc1'), (8192, 1024, 4096, False, False, 'roberta fc2'),
    (8192, 3072, 1024, False, False, 'roberta qkv'), (8192, 1024, 1024, False, False, 'roberta out / ctx K,V proj'),
    (512, 2048, 1024, False, False, 'dec linear1'), (512, 1024, 1024, False, False, 'dec linear2/q/out'),
    (512, 4096, 1024, False, False, 'dec fc1'), (512, 1024, 4096, False, False, 'dec fc2'),
    (512, 496, 1024, False, False, 'dec taps K=31'), (1024, 1024, 512, True, True, 'dW 1024x1024'),
    (4096, 1024, 512, True, True, 'dW fc1'), (1024, 1024, 8192, True, True, 'dW ctx K proj'),
    (512, 5002, 1024, True, False, 'head logits'), (50176, 64, 576, False, False, 'resnet l1 3x3'),
    (50176, 256, 64, False, False, 'resnet l1 1x1'), (3136, 1024, 256, False, False, 'resnet l3 1x1'),
    (784, 512, 4608, False, False, 'resnet l4 3x3'),
]
for M, N, K, of32, acc, label in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16()
    b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.zeros(M, N, device='cuda', dtype=torch.float32 if of32 else torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, out=out, accumulate=acc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(a, b, out=out, accumulate=acc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tile = 128 if ((M + 127) // 128) * ((N + 127) // 128) >= 256 else 64  # (label only)
    print('%-28s M%-6d N%-5d K%-5d tile%-3d %8.1f us %8.1f TF/s' % (label, M, N, K, tile, us, 2.0 * M * N * K / us / 1e6))
