#!/usr/bin/env python3
"""Phase stamps of pwsp_kernel (workgroup 0) for the small-map ops of the bench workload, plus event-timed launches back to back."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine, load_library
from yolo_puncture_amd.weights import synthetic_state
lib = load_library()
B, S = 32, 640
eng = Engine("s", 80, False, "bf16", 0, state=synthetic_state("s", 80, False))
eng.set_autotune(False)
im = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).cuda()
out = eng.forward(im)
torch.cuda.synchronize()
ops = eng.plan(B, S, S)
import itertools
for abl, (i, o) in itertools.product((0, 1, 2, 3, 4), enumerate(ops)):
    if not o["kernel"].startswith("pwsp") or (abl and not o["name"].startswith("model.8.")):
        continue
    lib.yp_debug_ablation(abl)
    for _ in range(3):
        eng.run_op(i, im, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.run_op(i, im, out)
    e1.record()
    torch.cuda.synchronize()
    v = (C.c_uint64 * 32)()
    lib.yp_debug_pwsp_clocks(v)
    t0 = v[0]
    ph = [int(v[k]) - int(t0) for k in range(1, 6)]
    ks = [int(v[8 + g]) - int(t0) for g in range(16) if v[8 + g] >= t0]
    print(f"abl {abl} {o['name']:30s} {o['kernel']:20s} {e0.elapsed_time(e1) * 100:.1f} us/launch   cycles: prologue {ph[0]} gemm {ph[1]} epilogue {ph[2]} barrier {ph[3]} spatial {ph[4]}   k-steps at {ks}")
