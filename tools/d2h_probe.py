import torch, time, os
x = torch.randn(17*1024*1024//4, device='cuda')
h = torch.empty_like(x, device='cpu').pin_memory()
s = torch.cuda.Stream()
for trial in range(3):
    torch.cuda.synchronize(); t0=time.time()
    h.copy_(x, non_blocking=True); torch.cuda.synchronize()
    print("D2H 17MB: %.3f ms -> %.1f GB/s" % ((time.time()-t0)*1e3, 17/1024/(time.time()-t0)))
y = torch.empty_like(x)
for trial in range(2):
    torch.cuda.synchronize(); t0=time.time()
    x.copy_(h, non_blocking=True); torch.cuda.synchronize()
    print("H2D 17MB: %.3f ms" % ((time.time()-t0)*1e3))
# interference: tiny kernels during the copy
a = torch.zeros(256, device='cuda')
def tiny(n=200):
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): a.add_(1)
    torch.cuda.synchronize(); return (time.time()-t0)*1e3/n
print("tiny alone %.4f ms" % tiny())
with torch.cuda.stream(s):
    for _ in range(20): h.copy_(x, non_blocking=True)
print("tiny with D2H in flight %.4f ms" % tiny())
torch.cuda.synchronize()
