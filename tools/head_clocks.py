#!/usr/bin/env python3
"""Print the phase breakdown of head_select_kernel (image 0's workgroup) for a bs-32 v10-S forward."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_puncture_amd.engine import Engine, load_library
from yolo_puncture_amd.weights import synthetic_state
lib = load_library()
eng = Engine("s", 80, False, "bf16", 0, state=synthetic_state("s", 80, False))
im = torch.randint(0, 256, (32, 640, 640, 3), dtype=torch.uint8).cuda()
for _ in range(3):
    eng.forward(im)
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)()
lib.yp_debug_head_clocks(buf)
t = [int(x) for x in buf]
names = ["load keys", "stage-1 bound T0", "stage-1 select", "stage-2 scan", "stage-2 select", "decode"]
for i, n in enumerate(names):
    print(f"{n:16s} {(t[i + 1] - t[i]) / 100.0:8.2f} us")
print(f"{'total':16s} {(t[6] - t[0]) / 100.0:8.2f} us")
print(f"stage-2 rounds {t[7] >> 32}, candidates in the last round's select {t[7] & 0xffffffff}")
buf = (C.c_uint64 * 8)()
lib.yp_debug_head_branch_clocks(buf)
t = [int(x) for x in buf]
names = ["position list", "plane 0", "channel chunks", "store"]
print(f"position kernel, workgroup 0's first tile (level {t[5] >> 32}, {t[5] & 0xffffffff} chunks): " + ", ".join(f"{n} {(t[i + 1] - t[i]) / 100.0:.2f}" for i, n in enumerate(names))
      + f" | total {(t[4] - t[0]) / 100.0:.2f} us; launch: {t[6]} tiles, {t[7]} positions")
