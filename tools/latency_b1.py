#!/usr/bin/env python3
"""Per-frame latency of the reference's real call shape: one frame per `.predict` (yolo_seg/app.py:85-91).

For YOLOv10-N and -S, frames of 720x1280 (-> letterbox 384x640) and 1080x810 (-> 640x480):
  engine_ms   yp_forward alone on a letterboxed frame resident in HBM, hipGraph replay (what the engine costs per frame)
  eager_ms    the same with eager launches (what round 1's predictor did)
  predict_ms  YOLO.predict(frame ndarray on the host): upload + device LetterBox + forward + conf filter + scale_boxes (+ D2H of boxes)
Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from yolo_puncture_amd import YOLO  # noqa: E402
from yolo_puncture_amd.engine import Engine  # noqa: E402
from yolo_puncture_amd.weights import synthetic_state  # noqa: E402


def timed(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    out = {}
    g = torch.Generator().manual_seed(0)
    for variant in ("n", "s"):
        eng = Engine(variant, 80, False, "bf16", 0, state=synthetic_state(variant, 80, False, seed=0))
        for (H, W), (h0, w0) in (((384, 640), (720, 1280)), ((640, 480), (1080, 810))):
            im = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8, generator=g).cuda()
            res = dict(det=torch.empty((1, 300, 6), device="cuda"), idx=torch.empty((1, 300), dtype=torch.int32, device="cuda"), coeff=None)
            eng.set_graph(False)
            eager = timed(lambda: eng.forward(im, res))
            eng.set_graph(True)
            graph = timed(lambda: eng.forward(im, res))
            model = YOLO(f"synthetic:{variant}")
            frame = np.random.default_rng(0).integers(0, 256, (h0, w0, 3), dtype=np.uint8)

            def pred():
                r = model.predict(frame, conf=0.25)[0]
                return r.boxes.cpu().numpy().xyxy

            p = timed(pred, n=100, warm=10)
            out[f"v10{variant}_1x{H}x{W}"] = dict(engine_ms=round(graph, 4), eager_ms=round(eager, 4), predict_ms=round(p, 4),
                                                   frame=f"{h0}x{w0}")
        eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
