#!/usr/bin/env python3
"""Run U^2-Net-P forward N times on a 380x380 crop (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.u2net import U2NetEngine, synthetic_state
dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
e = U2NetEngine("p", dt, 0, state=synthetic_state("p", 0))
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
e.set_graph(graph)
x = torch.randint(0, 256, (1, 380, 380, 3), dtype=torch.uint8).cuda()
import time
for _ in range(5):
    e.forward(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    e.forward(x)
torch.cuda.synchronize()
print(f"u2netp {dt} 380x380: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms / forward ({'hipGraph replay' if graph else 'eager'})")
e.close()
