"""What a plain streaming read sustains on this part (context for the HBM fractions): torch reductions / copies over buffers far larger than the Infinity Cache."""
import torch

dev = "cuda:0"
x = torch.empty(2 * 1024 ** 3, dtype=torch.float32, device=dev).normal_()   # 8 GB
y = torch.empty_like(x)


def timeit(f, n=10):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


t = timeit(lambda: x.sum())
print(f"sum over 8 GB fp32: {x.numel() * 4 / t / 1e12:.2f} TB/s read")
t = timeit(lambda: torch.max(x))
print(f"max over 8 GB fp32: {x.numel() * 4 / t / 1e12:.2f} TB/s read")
t = timeit(lambda: y.copy_(x))
print(f"copy 8 GB -> 8 GB: {2 * x.numel() * 4 / t / 1e12:.2f} TB/s read + write")
xv = x[: (x.numel() // 320) * 320].view(-1, 320)   # 1280-byte rows
del y
idx = torch.randint(0, xv.shape[0], (4_000_000,), device=dev)
out = torch.empty((idx.numel(), 320), dtype=torch.float32, device=dev)
t = timeit(lambda: torch.index_select(xv, 0, idx, out=out), n=3)
print(f"index_select of 4 M random 1280-byte rows: {idx.numel() * 1280 / t / 1e12:.2f} TB/s read (+ the same written)")
