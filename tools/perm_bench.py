import torch, time
n = 3200000
x = torch.randn(n, device="cuda"); p64 = torch.randperm(n, device="cuda"); p32 = p64.int()
def t(f, k=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
print("index_select i64 %.1f us" % t(lambda: x.index_select(0, p64)))
print("index_select i32 %.1f us" % t(lambda: x.index_select(0, p32)))
print("x[p64] %.1f us" % t(lambda: x[p64]))
print("gather i64 %.1f us" % t(lambda: torch.gather(x, 0, p64)))
print("take %.1f us" % t(lambda: torch.take(x, p64)))
print("clone %.1f us" % t(lambda: x.clone()))
