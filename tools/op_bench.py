#!/usr/bin/env python3
"""Time ONE op of the plan in a loop (optionally forcing a conv tile config); used under rocprofv3 --pmc."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine, load_library
from yolo_puncture_amd.weights import synthetic_state
ap = argparse.ArgumentParser()
ap.add_argument("--op", default="model.4.m.0.cv1"); ap.add_argument("--cfg", type=int, default=-1)
ap.add_argument("--iters", type=int, default=20); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--variant", default="s"); ap.add_argument("--imgsz", type=int, default=640)
ap.add_argument("--ablate", type=int, default=0)
ap.add_argument("--flush", default="", help="op to run before every timed run (evicts the caches), e.g. model.0")
a = ap.parse_args()
lib = load_library()
eng = Engine(a.variant, 80, False, "bf16", 0, state=synthetic_state(a.variant, 80, False))
eng.set_autotune(False)
im = torch.randint(0, 256, (a.batch, a.imgsz, a.imgsz, 3), dtype=torch.uint8).cuda()
out = eng.forward(im)
torch.cuda.synchronize()
ops = eng.plan(a.batch, a.imgsz, a.imgsz)
idx = [i for i, o in enumerate(ops) if o["name"] == a.op][0]
lib.yp_debug_force_conv_cfg(a.cfg)
lib.yp_debug_ablation(a.ablate)
fidx = [i for i, o in enumerate(ops) if o["name"] == a.flush][0] if a.flush else -1
for _ in range(3):
    eng.run_op(idx, im, out)
torch.cuda.synchronize()
tot = 0.0
for _ in range(a.iters):
    if fidx >= 0:
        eng.run_op(fidx, im, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.run_op(idx, im, out)
    e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1) * 1e-3
dt = tot / a.iters
o = ops[idx]
print(f"{a.op} cfg {a.cfg}: {dt*1e6:.1f} us  {o['flops']/dt/1e12:.1f} TF  {o['bytes']/dt/1e9:.0f} GB/s")
