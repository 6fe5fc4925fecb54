#!/usr/bin/env python3
"""Bounded stress of the eager per-op path (yp_profile: one launch + two event records per op) and of the graph replay
with fresh output tensors per call; run once on the GPU box before a round ends (host-side robustness check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state

eng = Engine("s", 80, False, "bf16", 0, state=synthetic_state("s", 80, False))
im = torch.randint(0, 256, (32, 640, 640, 3), dtype=torch.uint8).cuda()
t0 = time.time()
ref = None
eng.set_graph(True)
for i in range(300):                      # graph replay, new output tensors every call
    out = eng.forward(im)
    if i % 100 == 0:
        torch.cuda.synchronize()
        d = out["det"].clone()
        assert ref is None or torch.equal(d, ref), "replay is not deterministic"
        ref = d
        print("replay", i, round(time.time() - t0, 1), "s", flush=True)
torch.cuda.synchronize()
eng.set_graph(False)
for r in range(6):                        # eager: 6 x 20 iterations x ~95 launches
    ops = eng.profile(im, iters=20)
    print("eager round", r, "sum %.3f ms" % sum(o["ms"] for o in ops if o["kernel"] != "-"), round(time.time() - t0, 1), "s", flush=True)
eng.set_graph(True)
out = eng.forward(im)
torch.cuda.synchronize()
assert torch.equal(out["det"], ref)
print("stress ok")
