import torch, time
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for mb in (256, 755, 2048):
    x=torch.empty(mb*1024*1024//2, dtype=torch.float16, device="cuda"); y=torch.empty_like(x)
    ms=t(lambda: x.fill_(1.0)); print(f"fill {mb} MB: {ms:.3f} ms  {mb/1024/ms*1000:.0f} GB/s written")
    ms=t(lambda: y.copy_(x)); print(f"copy {mb} MB: {ms:.3f} ms  {2*mb/1024/ms*1000:.0f} GB/s (r+w)")
    ms=t(lambda: x.sum()); print(f"sum  {mb} MB: {ms:.3f} ms  {mb/1024/ms*1000:.0f} GB/s read")
