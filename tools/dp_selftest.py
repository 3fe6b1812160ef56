"""1-rank RCCL micro-timings (launch under torchrun --nproc-per-node 1): what the DP collectives cost by themselves."""
import os, time, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
x = torch.zeros(135_000_000, device='cuda')
s = torch.ones(1, device='cuda')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); host = (time.perf_counter() - t0) / n * 1e3; torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, host
def buckets():
    b = 256 * (1 << 20) // 4
    hs = [dist.all_reduce(x[i:i + b], async_op=True) for i in range(0, x.numel(), b)]
    for h in hs: h.wait()
print('all_reduce 540 MB in 3 buckets: gpu %.3f ms, host %.3f ms' % t(buckets))
print('all_reduce 1 scalar:            gpu %.3f ms, host %.3f ms' % t(lambda: dist.all_reduce(s)))
side = torch.cuda.Stream()
def on_side():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        buckets()
    torch.cuda.current_stream().wait_stream(side)
print('same on a side stream:          gpu %.3f ms, host %.3f ms' % t(on_side))
dist.destroy_process_group()
