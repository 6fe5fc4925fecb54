#!/usr/bin/env python3
"""Per-frame latency of the reference's real call shapes (one frame per call, yolo_seg/app.py:85-105,184):

  v10{n,s}_1xHxW   yp_forward on a letterboxed frame resident in HBM: hipGraph replay (pipelined and with a sync per frame), eager launches
                   (round 1's predictor), and YOLO.predict(frame on the host): upload + device LetterBox + forward + conf filter + scale_boxes
  {11n,v8n}-seg    YOLO.predict(frame, retina_masks=True) + masks.xy[best] + min_rect_len: the per-frame YOLO work of app.py:91-103,
                   contour and rectangle on the device
  u2netp_380       unet_predict on the 380x380 crop (app.py:184): fp32 (parity mode) and bf16 engines; the oracle (torch-CPU, the
                   reference's own arithmetic) timed beside it on this box's host cores
Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from yolo_puncture_amd import YOLO  # noqa: E402
from yolo_puncture_amd.engine import Engine  # noqa: E402
from yolo_puncture_amd.u2net import U2NetEngine, synthetic_state as u2_state  # noqa: E402
from yolo_puncture_amd.weights import synthetic_state  # noqa: E402


def timed(fn, n=200, warm=20, sync_each=False):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each:
            torch.cuda.synchronize()
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
            eager_sync = timed(lambda: eng.forward(im, res), sync_each=True)
            eng.set_graph(True)
            graph = timed(lambda: eng.forward(im, res))
            graph_sync = timed(lambda: eng.forward(im, res), sync_each=True)
            model = YOLO(f"synthetic:{variant}")
            frame = np.random.default_rng(0).integers(0, 256, (h0, w0, 3), dtype=np.uint8)

            def pred():
                r = model.predict(frame, conf=0.25)[0]
                return r.boxes.cpu().numpy().xyxy

            p = timed(pred, n=100, warm=10)
            out[f"v10{variant}_1x{H}x{W}"] = dict(graph_ms=round(graph, 4), graph_sync_ms=round(graph_sync, 4), eager_ms=round(eager, 4),
                                                   eager_sync_ms=round(eager_sync, 4), predict_ms=round(p, 4), frame=f"{h0}x{w0}")
        eng.close()
    # the families the app's UI offers, with the per-frame tail of app.py:91-103
    from helpers import make_case_family
    from yolo_puncture_amd.weights import save_as_ultralytics_pt
    import tempfile
    for fam in ("11", "v8"):
        st, ims = make_case_family(fam, "n", 80, 0, (1, 384, 640))
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "m.pt")
            save_as_ultralytics_pt(st, path)
            model = YOLO(path)
            frame = np.ascontiguousarray(np.repeat(np.repeat(ims[0].numpy(), 2, 0), 2, 1)[:720, :1280])

            # a confidence that leaves a video-like number of detections (the synthetic net fires on everything at 0.25)
            scores = np.sort(model.predict(frame, conf=0.01)[0].boxes.cpu().numpy().conf)[::-1]
            conf = float(scores[min(7, len(scores) - 1)]) - 1e-6 if len(scores) else 0.25

            def step():
                r = model.predict(frame, conf=conf, retina_masks=True)[0]
                b = r.boxes.cpu().numpy()
                if len(b.cls):
                    best = int(np.argmax(b.conf))
                    _ = r.masks.xy[best]
                    return r.masks.min_rect_len(best)
                return None

            out[f"{fam}n-seg_predict+xy+rect_720p"] = dict(ms=round(timed(step, n=60, warm=6), 4), detections=len(model.predict(frame, conf=conf)[0].boxes))
    # U^2-Net-P on the 380x380 crop
    st = u2_state("p", 0)
    crop = torch.randint(0, 256, (1, 380, 380, 3), dtype=torch.uint8, generator=g).cuda()
    for dt in ("fp32", "bf16"):
        e = U2NetEngine("p", dt, 0, state=st)
        out[f"u2netp_380_{dt}"] = dict(ms=round(timed(lambda: e.forward(crop), n=100, warm=10), 4),
                                       sync_ms=round(timed(lambda: e.forward(crop), n=100, warm=10, sync_each=True), 4))
        e.close()
    from oracle.u2net_oracle import U2NetOracle
    x = crop.cpu().flip(-1).permute(0, 3, 1, 2).float() / 255.0
    cores = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    torch.set_num_threads(min(cores, 32))
    o = U2NetOracle(st, "p")
    with torch.no_grad():
        o.forward(x)
        t0 = time.perf_counter()
        for _ in range(3):
            o.forward(x)
        cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    out["u2netp_380_cpu_oracle"] = dict(ms=round(cpu_ms, 2), cores=torch.get_num_threads())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
