#!/usr/bin/env python3
"""Per-op HIP-event timing table of the engine's plan (runs on the GPU box): op, kernel, ms, TFLOP/s, GB/s."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="s"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--imgsz", type=int, default=640); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--seg", action="store_true"); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--out", default="")
a = ap.parse_args()
eng = Engine(a.variant, 80, a.seg, a.dtype, 0, state=synthetic_state(a.variant, 80, a.seg))
im = torch.randint(0, 256, (a.batch, a.imgsz, a.imgsz, 3), dtype=torch.uint8).cuda()
eng.profile(im, iters=2)
ops = eng.profile(im, iters=a.iters)
tot = sum(o["ms"] for o in ops if o["kernel"] != "-")
lines = []
for o in ops:
    if o["kernel"] == "-":
        lines.append(f'{o["name"]:34s} (fused into a later op: no launch)')
        continue
    ms = max(o["ms"], 1e-6)
    lines.append(f'{o["name"]:34s} {o["kernel"][:38]:38s} {o["ms"]*1e3:8.1f} us {o["flops"]/ms/1e9:8.1f} TF {o["bytes"]/ms/1e6:8.0f} GB/s  {o["flops"]/1e9:7.2f} GF {o["bytes"]/1e6:7.1f} MB')
lines.append(f"total {tot:.3f} ms  ->  {a.batch/tot*1e3:.0f} img/s (eager, event-timed sum)")
txt = "\n".join(lines)
print(txt)
if a.out:
    with open(a.out, "w") as f:
        f.write(txt + "\n")
