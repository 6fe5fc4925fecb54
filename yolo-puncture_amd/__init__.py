"""yolo-puncture_amd: MI355X-native YOLOv10 predict path behind the reference's `YOLO(...).predict(...)` surface.

Hot path = hand-written HIP (gfx950) behind the C-ABI in include/yolop.h (csrc/). This package is the thin
Python host: weight loading, the ctypes binding, and the `YOLO`/`Results` facade the reference's callers use
(yolo_seg/app.py:45-101, yolo_seg/yolo_with_deva.py:37-88).
"""
__version__ = "0.1.0"

def __getattr__(name):
    # lazy: importing the package must not require the built .so (tests of pure host logic run without it)
    if name in ("YOLO", "Results", "Boxes", "Masks"):
        from . import predictor
        return getattr(predictor, name)
    if name in ("Engine", "load_library"):
        from . import engine
        return getattr(engine, name)
    if name in ("load_unet", "unet_predict", "U2NetEngine"):
        from . import u2net
        return getattr(u2net, name)
    if name == "auto_segment":
        from .deva_adapter import auto_segment
        return auto_segment
    raise AttributeError(name)
