#!/usr/bin/env python3
"""Probe: throughput of D engines (same weights) fed round-robin from D caller streams, i.e. D steps in flight."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state
ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=2); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
st = synthetic_state("s", 80, False, seed=0)
dev = torch.device("cuda", 0)
frames = torch.randint(0, 256, (a.batch, 640, 640, 3), dtype=torch.uint8).to(dev)
engs = [Engine("s", 80, False, "bf16", 0, state=st) for _ in range(a.depth)]
outs = [dict(det=torch.empty((a.batch, 300, 6), device=dev), idx=torch.empty((a.batch, 300), dtype=torch.int32, device=dev), coeff=None) for _ in range(a.depth)]
streams = [torch.cuda.Stream(dev) for _ in range(a.depth)]
for e in engs:
    e.forward(frames, outs[0]); torch.cuda.synchronize(); e.set_graph(True)
def run(n):
    for i in range(n):
        k = i % a.depth
        with torch.cuda.stream(streams[k]):
            engs[k].forward(frames, outs[k])
run(2 * a.depth + 4); torch.cuda.synchronize()
t0 = time.perf_counter(); run(a.steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"depth {a.depth}: {dt / a.steps * 1e3:.4f} ms/step, {a.batch * a.steps / dt:.1f} img/s")
ref = outs[0]["det"].clone()
print("all engines agree:", all(torch.equal(o["det"], ref) for o in outs))
