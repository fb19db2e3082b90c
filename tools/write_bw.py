import torch
x = torch.empty(64 * 1024 * 1024 // 2, dtype=torch.bfloat16, device="cuda")
big = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device="cuda")
for name, t in (("64 MB", x), ("512 MB", big)):
    for _ in range(3): t.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): t.zero_()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(name, "fill:", round(us, 1), "us", round(t.numel() * t.element_size() / us / 1e6, 2), "TB/s")
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y.copy_(x)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print("64 MB copy:", round(us, 1), "us", round(2 * 64 / us, 2), "TB/s (read + write)")
