import torch, time
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
b = torch.empty_like(a)
for name, fn, bytes_ in (("copy (1R+1W)", lambda: b.copy_(a), 2 * a.numel() * 4),
                         ("fill (1W)", lambda: a.fill_(1.0), a.numel() * 4),
                         ("sum (1R)", lambda: a.sum(), a.numel() * 4),
                         ("add 2R+1W", lambda: torch.add(a, b, out=b), 3 * a.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%-14s %.3f ms  %.2f TB/s" % (name, dt * 1e3, bytes_ / dt / 1e12))
