import torch
for M, N, K in ((16384, 4096, 1024), (16384, 1024, 4096), (16384, 3072, 1024), (16384, 1024, 1024)):
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(a, b.t(), out=out)
torch.cuda.synchronize()
