"""Two batches in flight: how the step time depends on which torch streams carry the two engines (the runtime maps streams onto a few
hardware queues; two graphs on ONE queue cannot overlap). Same box, same engines, graphs as one chain each (yp_set_graph 2).
    python tools/ring_queue_probe.py [--steps 60]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, S = 32, 640
st = synthetic_state("s", 80, False, seed=0)
frames = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).to(dev)
engs, cfgs = [], None
for i in range(3):
    e = Engine("s", 80, False, "bf16", 0, state=st)
    if cfgs is not None:
        e.tuning_import(B, S, S, cfgs)
    e.forward(frames)
    torch.cuda.synchronize()
    if cfgs is None:
        cfgs = e.tuning_export()
    engs.append(e)
outs = [dict(det=torch.empty((B, 300, 6), device=dev), idx=torch.empty((B, 300), dtype=torch.int32, device=dev), coeff=None) for _ in range(3)]


def measure(tag, streams, mode):
    for e in engs:
        e.set_graph(mode)
    def run(k):
        for i in range(k):
            j = i % len(streams)
            with torch.cuda.stream(streams[j]):
                engs[j].forward(frames, outs[j])
    run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{tag:58s} {dt / a.steps * 1e3:.4f} ms/step", flush=True)


s_def = [torch.cuda.Stream(dev) for _ in range(6)]
s_hi = [torch.cuda.Stream(dev, priority=-1) for _ in range(3)]
measure("one in flight, graph with lanes", [s_def[0]], 1)
measure("one in flight, chain", [s_def[0]], 2)
for k in range(1, 6):
    measure(f"two in flight, chains, default-priority streams 0 and {k}", [s_def[0], s_def[k]], 2)
measure("two in flight, chains, default + high priority", [s_def[0], s_hi[0]], 2)
measure("two in flight, chains, high + high priority", [s_hi[0], s_hi[1]], 2)
measure("two in flight, lanes, default + high priority", [s_def[0], s_hi[0]], 1)
measure("two in flight, lanes, default 0 and 1", [s_def[0], s_def[1]], 1)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for k in (1, 2, 3, 4):
    measure(f"three in flight, chains, default 0 + high 0 + default {k}", [s_def[0], s_hi[0], s_def[k]], 2)
measure("three in flight, chains, default 0 + high 0 + high 1", [s_def[0], s_hi[0], s_hi[1]], 2)
measure("three in flight, chains, default 0, 2, 3", [s_def[0], s_def[2], s_def[3]], 2)
measure("two in flight again, default + high", [s_def[0], s_hi[0]], 2)
