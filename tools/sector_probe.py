"""Does HBM traffic come in 32-byte sectors?  Read the first 8 / 16 floats of every 16-float (64 B)
row of a 1 GiB array; if the half-row read takes about half the time, fetches are sector-granular.
Measured (MI355X): 0.265 ms per half against 0.419 ms whole, and under `rocprofv3 --pmc FETCH_SIZE` the
half-row kernels fetch the full 1 GiB: fetches are whole 64-byte (or larger) lines, the time saved is
the halved write."""
import torch, time
n = 1 << 24                                     # rows of 64 B -> 1 GiB
a = torch.rand(n, 16, device="cuda")
out = torch.empty(n, 8, device="cuda")
full = torch.empty(n, 16, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
t_half = t(lambda: out.copy_(a[:, :8]))         # reads 32 B of each 64 B row, writes 32 B
t_full = t(lambda: full.copy_(a))               # reads 64 B, writes 64 B
t_half2 = t(lambda: out.copy_(a[:, 8:]))
print("copy first half of each 64-B row: %.3f ms; second half: %.3f ms; whole rows: %.3f ms" % (t_half * 1e3, t_half2 * 1e3, t_full * 1e3))
print("bytes if sector-granular: half = 0.5 GiB read + 0.5 written; if 64-B granular: 1 GiB read + 0.5 written")
