"""A/B probe: N batches in flight. N engines (own arena, own graph, own stream; same weights) take the steps round-robin, each
on its own torch stream, so that the low-occupancy parts of one batch's graph (the 20x20 tail, the top-k kernel: 100-400
workgroups on 256 CUs) run beside another batch's kernels. Prints images/s for N = 1, 2, 3 on the same box. (The probe that led to
parallel.EngineRing; its graphs keep their head lanes, so its numbers depend on the runtime's stream -> hardware-queue mapping - DESIGN 4,
round 3. bench.py --in-flight N measures the ring itself.)
    python tools/inflight_probe.py [--steps 40]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yolo_puncture_amd.engine import Engine  # noqa: E402
from yolo_puncture_amd.weights import synthetic_state  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--max-inflight", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, S = a.batch, 640
st = synthetic_state("s", 80, False, seed=0)
frames = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).to(dev)
engs, cfgs = [], None
for i in range(a.max_inflight):
    e = Engine("s", 80, False, "bf16", 0, state=st)
    if cfgs is not None:
        e.tuning_import(B, S, S, cfgs)          # every engine runs the same tile configurations
    e.forward(frames)
    torch.cuda.synchronize()
    if cfgs is None:
        cfgs = e.tuning_export()
    e.set_graph(True)
    e.forward(frames)
    engs.append(e)
torch.cuda.synchronize()
for n in list(range(1, a.max_inflight + 1)) + [1]:
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    outs = [dict(det=torch.empty((B, 300, 6), device=dev), idx=torch.empty((B, 300), dtype=torch.int32, device=dev), coeff=None) for _ in range(n)]
    def run(k):
        for i in range(k):
            j = i % n
            with torch.cuda.stream(streams[j]):
                engs[j].forward(frames, outs[j])
    run(2 * n + 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in flight {n}: {B * a.steps / dt:9.1f} img/s   {dt / a.steps * 1e3:.4f} ms/step", flush=True)
