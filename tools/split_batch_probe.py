#!/usr/bin/env python3
"""Probe: is one 32-frame forward slower than two concurrent 16-frame forwards (two engines, two hipGraphs on their own streams)?
The 20x20 / 40x40 layers under-fill 256 CUs; a second half-batch in flight could fill the gaps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state
st = synthetic_state("s", 80, False, seed=0)
def run(nsplit, B=32, steps=40):
    engs = [Engine("s", 80, False, "bf16", 0, state=st) for _ in range(nsplit)]
    ims = [torch.randint(0, 256, (B // nsplit, 640, 640, 3), dtype=torch.uint8).cuda() for _ in range(nsplit)]
    outs = [dict(det=torch.empty((B // nsplit, 300, 6), device="cuda"), idx=torch.empty((B // nsplit, 300), dtype=torch.int32, device="cuda"), coeff=None) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    for e in engs: e.set_graph(True)
    def step():
        for e, im, o, s in zip(engs, ims, outs, streams):
            with torch.cuda.stream(s):
                e.forward(im, o)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for e in engs: e.close()
    return dt * 1e3
for n in (1, 2, 4, 1, 2):
    ms = run(n)
    print(f"{n} x {32 // n} frames concurrently: {ms:.3f} ms per 32 frames = {32 / ms * 1e3:.0f} img/s", flush=True)
