import sys, torch
sys.path.insert(0, '.')
import tell_amd
from tell_amd import ops
M, N, K = 8192, 4096, 1024            # RoBERTa fc1 of the bench workload (B=16, S=512)
a = torch.randn(M, K, device='cuda').bfloat16()
b = torch.randn(N, K, device='cuda').bfloat16()
bias = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(a, b, out=out, bias=bias, bias_mode=1, act=2)
torch.cuda.synchronize()
