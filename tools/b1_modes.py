"""One 384x640 frame per call (the reference's call shape, yolo_seg/app.py:85-91): yp_forward in its three launch modes, same box.
    eager | hipGraph with head lanes | hipGraph as one linear chain"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state


def timed(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for variant in ("n", "s"):
    eng = Engine(variant, 80, False, "bf16", 0, state=synthetic_state(variant, 80, False, seed=0))
    im = torch.randint(0, 256, (1, 384, 640, 3), dtype=torch.uint8).cuda()
    res = dict(det=torch.empty((1, 300, 6), device="cuda"), idx=torch.empty((1, 300), dtype=torch.int32, device="cuda"), coeff=None)
    eng.forward(im, res)
    nk = sum(1 for o in eng.plan(1, 384, 640) if o["kernel"] != "-")
    row = {}
    for name, mode in (("eager", 0), ("graph+lanes", 1), ("graph linear", 2), ("auto", 3)):
        eng._chk(eng.lib.yp_set_graph(eng._h, mode))
        row[name] = timed(lambda: eng.forward(im, res))
    print(f"v10-{variant}: {nk} launches; " + "  ".join(f"{k} {v:.4f} ms" for k, v in row.items()), flush=True)
    eng.close()
