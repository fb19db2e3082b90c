import torch, time
dev = "cuda"
def t(fn, n=20):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (16, 32, 64, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    # write y then read it back (producer -> consumer), per iteration: mul (r x, w y) + sum (r y)
    def both():
        torch.mul(x, 2.0, out=y); y.sum()
    def mul_only():
        torch.mul(x, 2.0, out=y)
    def sum_only():
        y.sum()
    tm, ts, tb = t(mul_only), t(sum_only), t(both)
    print(f"{mb:5d} MB: mul {2*mb/1e3/tm/1e3*1e3:7.2f} GB/s-ish  mul {tm*1e6:8.1f} us ({2*mb*1.048576/tm/1e6:6.2f} TB/s)  sum {ts*1e6:8.1f} us ({mb*1.048576/ts/1e6:6.2f} TB/s)  both {tb*1e6:8.1f} us")
