#!/usr/bin/env python3
"""Phase clocks of yp_mask_contours for the masks.xy[best] call of the app loop on the synthetic v8n-seg / 11n-seg nets (720p frame),
for both strategies:  python tools/contour_probe.py"""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_case_family
from yolo_puncture_amd import YOLO
from yolo_puncture_amd import predictor
from yolo_puncture_amd.engine import load_library
from yolo_puncture_amd.weights import save_as_ultralytics_pt
lib = load_library()
names = ["box", "bit image", "candidates", "trace+select", "emit", "hull+rect"]
for fam in ("v8", "11"):
    st, ims = make_case_family(fam, "n", 80, 0, (1, 384, 640))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.pt")
        save_as_ultralytics_pt(st, path)
        model = YOLO(path)
        frame = np.ascontiguousarray(np.repeat(np.repeat(ims[0].numpy(), 2, 0), 2, 1)[:720, :1280])
        scores = np.sort(model.predict(frame, conf=0.01)[0].boxes.cpu().numpy().conf)[::-1]
        conf = float(scores[min(7, len(scores) - 1)]) - 1e-6 if len(scores) else 0.25
        for strat in ("all", "largest"):
            predictor.MASK_POLYGON_STRATEGY = strat
            for rep in range(3):
                r = model.predict(frame, conf=conf, retina_masks=True)[0]
                b = r.boxes.cpu().numpy()
                best = int(np.argmax(b.conf))
                torch.cuda.synchronize(); t0 = time.perf_counter()
                xy = r.masks.xy[best]
                rl = r.masks.min_rect_len(best)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            buf = (C.c_uint64 * 12)()
            lib.yp_debug_contour_clocks(buf)
            t = [int(x) for x in buf]
            print(f"{fam}n-seg {strat:8s}: xy+rect {dt:.3f} ms; polygon {len(xy)} points; candidates {t[8]}, points {t[9]}, box {t[10]}x{t[11]}; "
                  + ", ".join(f"{n} {(t[i + 1] - t[i]) / 100.0:.1f}" for i, n in enumerate(names)) + " us")
