#!/usr/bin/env python3
"""The bench workload (YOLOv10-S 640x640 bs 32, bench.py's weights and frames) as eager launches with a marker kernel in front of every
op: run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two passes; tools/refresh_profiles.sh), then tools/op_traffic.py turns
the per-dispatch counter rows into HBM bytes per OP NAME. Writes the op order to gpurun_out/refresh/op_order.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine, _stream_ptr
from yolo_puncture_amd.weights import synthetic_state

B, S, REPS = 32, 640, 3
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "refresh", "op_order.json")
st = synthetic_state("s", 80, False, seed=0)
eng = Engine("s", 80, False, "bf16", 0, state=st)
g = torch.Generator().manual_seed(0)
frames = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).cuda()
out = eng.forward(frames)                 # plans, tunes (or reads YOLOP_TUNE_CACHE), one eager pass
torch.cuda.synchronize()
ops = eng.plan(B, S, S)
dev = frames.device
for rep in range(REPS):
    for i, o in enumerate(ops):
        eng._chk(eng.lib.yp_debug_marker(_stream_ptr(dev)))
        if o["kernel"] != "-":
            eng.run_op(i, frames, out)
eng._chk(eng.lib.yp_debug_marker(_stream_ptr(dev)))
torch.cuda.synchronize()
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump({"ops": [o["name"] for o in ops], "kernels": [o["kernel"] for o in ops], "reps": REPS}, open(out_path, "w"))
eng.close()
print("op_traffic_run: ok", len(ops), "ops x", REPS)
