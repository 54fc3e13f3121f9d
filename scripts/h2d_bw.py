import torch, time
x = torch.empty(512 << 20, dtype=torch.uint8).pin_memory()
y = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for _ in range(2): y.copy_(x, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t = time.perf_counter() - t
print("H2D pinned 1D: %.1f GB/s" % (5 * x.numel() / t / 1e9))
