"""One 384x640 frame per call, eager launches: is the 0.48 ms per call host enqueue time or GPU time?
host = wall time until the last yp_forward of a burst has RETURNED (queue still draining), total = until the GPU is idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state
eng = Engine("n", 80, False, "bf16", 0, state=synthetic_state("n", 80, False, seed=0))
im = torch.randint(0, 256, (1, 384, 640, 3), dtype=torch.uint8).cuda()
res = dict(det=torch.empty((1, 300, 6), device="cuda"), idx=torch.empty((1, 300), dtype=torch.int32, device="cuda"), coeff=None)
for _ in range(30):
    eng.forward(im, res)
torch.cuda.synchronize()
for n in (1, 4, 16, 64, 256):
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward(im, res)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"burst of {n:4d}: host {1e3 * (t1 - t0) / n:.4f} ms per call, total {1e3 * (t2 - t0) / n:.4f} ms per call, drain after the last return {1e3 * (t2 - t1):.3f} ms", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.forward(im, res); e1.record(); torch.cuda.synchronize()
print(f"one call between two events: {e0.elapsed_time(e1):.4f} ms on the GPU")
