import time, numpy as np, torch
f = np.random.randint(0, 255, (720, 1280, 3), dtype=np.uint8)
dev = torch.device("cuda", 0)
torch.cuda.synchronize()
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("pageable numpy -> cuda (720p u8, 2.76 MB): %.3f ms" % t(lambda: torch.from_numpy(f).to(dev, non_blocking=True)))
pin = torch.empty((720, 1280, 3), dtype=torch.uint8).pin_memory()
dst = torch.empty((720, 1280, 3), dtype=torch.uint8, device=dev)
def viapin():
    pin.numpy()[...] = f
    dst.copy_(pin, non_blocking=True)
print("numpy -> pinned staging -> cuda: %.3f ms" % t(viapin))
print("pinned -> cuda only: %.3f ms" % t(lambda: dst.copy_(pin, non_blocking=True)))
d = torch.empty((300, 6), device=dev)
print("D2H of [300,6] rows (.cpu()): %.3f ms" % t(lambda: d.cpu()))
