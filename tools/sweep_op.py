#!/usr/bin/env python3
"""Time one or more conv ops under every valid tile configuration (forced), to study what limits them."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine, load_library
from yolo_puncture_amd.weights import synthetic_state
ap = argparse.ArgumentParser()
ap.add_argument("--ops", default="model.8.cv1,model.2.cv2,model.16.cv1,model.6.m.0.cv2,model.1,model.4.m.0.cv1")
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--out", default=""); ap.add_argument("--cfgs", default="")
a = ap.parse_args()
lib = load_library()
eng = Engine("s", 80, False, "bf16", 0, state=synthetic_state("s", 80, False))
eng.set_autotune(False)
im = torch.randint(0, 256, (a.batch, 640, 640, 3), dtype=torch.uint8).cuda()
out = eng.forward(im); torch.cuda.synchronize()
ops = eng.plan(a.batch, 640, 640)
lines = []
cfgs = list(range(14)) + [100, 101, 102, 103] + [200, 201, 202, 203, 204] + list(range(300, 341)) + list(range(400, 409)) + list(range(500, 505)) + [600, 601] + list(range(700, 713)) + list(range(800, 808))
if a.cfgs: cfgs = [int(x) for x in a.cfgs.split(',')]
for name in a.ops.split(","):
    idx = [i for i, o in enumerate(ops) if o["name"] == name][0]
    o = ops[idx]
    for c in cfgs:
        lib.yp_debug_force_conv_cfg(c)
        kn = eng.plan(a.batch, 640, 640)[idx]["kernel"]   # plan is cached: kernel name not refreshed; informational only
        for _ in range(3): eng.run_op(idx, im, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): eng.run_op(idx, im, out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        lines.append(f"{name:22s} cfg {c:4d}  {us:8.1f} us  {o['flops']/us/1e6:7.1f} TF  {o['bytes']/us/1e3:7.0f} GB/s")
lib.yp_debug_force_conv_cfg(-1)
txt = "\n".join(lines)
print(txt)
if a.out: open(a.out, "w").write(txt + "\n")
