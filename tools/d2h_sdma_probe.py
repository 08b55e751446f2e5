import torch, time
x = torch.empty(300*1024*1024//8, dtype=torch.float64, device="cuda").normal_()
h = torch.empty(x.shape, dtype=torch.float64).pin_memory()
s = torch.cuda.Stream()
for _ in range(2):
    with torch.cuda.stream(s):
        h.copy_(x, non_blocking=True)
    s.synchronize()
t0=time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s):
        h.copy_(x, non_blocking=True)
    s.synchronize()
dt=(time.perf_counter()-t0)/5
print("D2H 300 MiB: %.2f ms  %.1f GB/s" % (dt*1e3, 0.3146/dt))
