#!/usr/bin/env python3
"""The reference's per-frame tail (app.py:91-103) on a yolov8n-seg layout, N times: predict(retina_masks=True) -> masks.xy[best] ->
min_rect_len(best). Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split; prints the wall time per frame and a host
phase split (predict / xy / rect)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_case_family
from yolo_puncture_amd.weights import save_as_ultralytics_pt
from yolo_puncture_amd.predictor import YOLO
fam = sys.argv[1] if len(sys.argv) > 1 else "v8"
st, ims = make_case_family(fam, "n", 80, 0, (1, 384, 640))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.pt")
    save_as_ultralytics_pt(st, path)
    model = YOLO(path)
    frame = np.ascontiguousarray(np.repeat(np.repeat(ims[0].numpy(), 2, 0), 2, 1)[:720, :1280])
    scores = np.sort(model.predict(frame, conf=0.01)[0].boxes.cpu().numpy().conf)[::-1]
    conf = float(scores[min(7, len(scores) - 1)]) - 1e-6 if len(scores) else 0.25
    t = [0.0, 0.0, 0.0]
    N = 60
    for it in range(N + 10):
        if it == 10:
            torch.cuda.synchronize(); t = [0.0, 0.0, 0.0]; t00 = time.perf_counter()
        a = time.perf_counter()
        r = model.predict(frame, conf=conf, retina_masks=True)[0]
        b = r.boxes.cpu().numpy()
        c0 = time.perf_counter()
        best = int(np.argmax(b.conf))
        _ = r.masks.xy[best]
        c1 = time.perf_counter()
        _ = r.masks.min_rect_len(best)
        c2 = time.perf_counter()
        t[0] += c0 - a; t[1] += c1 - c0; t[2] += c2 - c1
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t00) / N * 1e3
    print(f"{fam}n-seg 720p frame: {tot:.3f} ms/frame; predict+boxes {t[0] / N * 1e3:.3f}, masks.xy {t[1] / N * 1e3:.3f}, min_rect_len {t[2] / N * 1e3:.3f}; detections {len(b.conf)}")
    import ctypes as C
    from yolo_puncture_amd.engine import load_library
    buf = (C.c_uint64 * 12)()
    load_library().yp_debug_contour_clocks(buf)
    t = [int(x) for x in buf]
    names = ["bounding box", "bit image", "candidates", "trace all", "emit winner", "hull + calipers"]
    print("contour_kernel (last call): " + ", ".join(f"{n} {(t[i + 1] - t[i]) / 100.0:.1f} us" for i, n in enumerate(names)) +
          f"; candidates {t[8]}, winner points {t[9]}, box {t[10]}x{t[11]}")
