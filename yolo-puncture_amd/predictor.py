"""`YOLO(...).predict(...)` facade over the MI355X engine: the exact Python surface the reference's callers use.

Call sites mirrored (reference = /root/reference):
    YOLO(path)                                         yolo_seg/app.py:45, yolo_seg/yolo_with_deva.py:226,
                                                       dev_tools/auto_speed_calc.py:40, dev_tools/classify/cls_bbox_dataset_generate.py:66
    .predict(source=, conf=, retina_masks=, device=)   yolo_seg/app.py:49,91 ; yolo_seg/yolo_with_deva.py:51 (positional source) ;
                                                       dev_tools/auto_speed_calc.py:62 ; dev_tools/classify/cls_bbox_dataset_generate.py:48
    results[0].boxes.cpu().numpy() / .xyxy .conf .cls  yolo_seg/app.py:92-98 ; .xywhn cls_bbox_dataset_generate.py:52
    results[0].masks.xy[i] / .data[i] / len(masks)     yolo_seg/app.py:50,101 ; yolo_seg/yolo_with_deva.py:61-68
    yolo.model.parameters() / yolo.model.to(device)    yolo_seg/yolo_with_deva.py:42,130
Semantics follow SURVEY.md Appendix A.5-A.7 [U] (ultralytics is not vendored in the reference).
The network itself runs only through libyolop.so (engine.py); there is no CPU fallback.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import hostops
from .engine import Engine, letterbox_device, mask_contours_device
from .weights import read_ultralytics_pt, synthetic_state

_ENGINE_CACHE: Dict[tuple, Engine] = {}


def shutdown() -> None:
    """Destroy every cached engine now (device memory, streams, graphs). Registered with atexit so that it runs while the HIP
    runtime and torch are still alive - not from `Engine.__del__` during interpreter teardown."""
    while _ENGINE_CACHE:
        _, eng = _ENGINE_CACHE.popitem()
        try:
            eng.close()
        except Exception:
            pass


import atexit  # noqa: E402

atexit.register(shutdown)


class Boxes:
    """[n,6] rows = x1,y1,x2,y2 (original-image pixels), conf, cls; sorted by conf descending."""

    def __init__(self, data, orig_shape: Tuple[int, int]):
        self.data = data
        self.orig_shape = tuple(orig_shape)

    def _new(self, data):
        return Boxes(data, self.orig_shape)

    def cpu(self):
        return self._new(self.data.cpu() if isinstance(self.data, torch.Tensor) else self.data)

    def numpy(self):
        return self._new(self.data.detach().cpu().numpy() if isinstance(self.data, torch.Tensor) else self.data)

    def to(self, *a, **k):
        return self._new(self.data.to(*a, **k) if isinstance(self.data, torch.Tensor) else self.data)

    def __len__(self):
        return int(self.data.shape[0])

    def __getitem__(self, i):
        d = self.data[i]
        return self._new(d.reshape(-1, 6) if d.ndim == 1 else d)

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, 4]

    @property
    def cls(self):
        return self.data[:, 5]

    @property
    def xywh(self):
        x = self.xyxy
        cat = torch.stack if isinstance(x, torch.Tensor) else np.stack
        return cat(((x[:, 0] + x[:, 2]) / 2, (x[:, 1] + x[:, 3]) / 2, x[:, 2] - x[:, 0], x[:, 3] - x[:, 1]), -1)

    @property
    def xyxyn(self):
        h, w = self.orig_shape
        x = self.xyxy
        s = x.new_tensor([w, h, w, h]) if isinstance(x, torch.Tensor) else np.asarray([w, h, w, h], dtype=x.dtype)
        return x / s

    @property
    def xywhn(self):
        h, w = self.orig_shape
        x = self.xywh
        s = x.new_tensor([w, h, w, h]) if isinstance(x, torch.Tensor) else np.asarray([w, h, w, h], dtype=x.dtype)
        return x / s


MASK_POLYGON_STRATEGY = "all"     # ultralytics masks2segments(strategy=...) [U]: "all" is the default of the 8.3.x line the app's YOLO11 weights
                                  # need - every external contour, laid end to end (8.1 ... early 8.3) or, "all_merged", bridged at their closest
                                  # points with merge_multi_segment (later 8.3.x); "largest" is the default of 8.0. All three give the same
                                  # minimum-area rectangle whenever a mask has one contour, and "all" / "all_merged" always (same point set).
                                  # SURVEY A.7; set before predicting.


class Masks:
    """.data: float {0,1} [n,H,W] (H,W = original image when retina_masks else the letterboxed input);
    .xy: one float32 [m,2] polygon per mask - cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + masks2segments(MASK_POLYGON_STRATEGY)
    as hostops defines them, in pixels of the original image.
    When the masks live on the GPU the contour, its convex hull and the minimum-area rectangle are computed there
    (yp_mask_contours) and only the polygon's few hundred points travel to the host; `.min_rect` is the rectangle
    `get_coord_min_rect_len` would derive from `.xy[i]` (reference yolo_seg/app.py:101-103)."""

    def __init__(self, data: Optional[torch.Tensor], orig_shape: Tuple[int, int], u8: Optional[torch.Tensor] = None, strategy: Optional[str] = None):
        if data is None and u8 is None:
            raise ValueError("Masks needs data or u8")
        self.strategy = strategy or MASK_POLYGON_STRATEGY
        self._data = data                 # float {0,1}; made from the uint8 masks on first use (the engine produces uint8)
        self.orig_shape = tuple(orig_shape)
        self._u8 = u8                     # the engine's uint8 masks (same pixels as data), kept for the device contour pass
        self._polys: Dict[int, np.ndarray] = {}
        self._rects: Dict[int, Tuple[float, float]] = {}

    @property
    def data(self):
        if self._data is None:
            self._data = self._u8.to(torch.float32)
        return self._data

    @property
    def shape(self):
        return tuple((self._data if self._data is not None else self._u8).shape)

    def cpu(self):
        m = Masks(self.data.cpu(), self.orig_shape, self._u8, self.strategy)
        m._polys, m._rects = self._polys, self._rects
        return m

    def numpy(self):
        d = self.data
        m = Masks(d.detach().cpu().numpy() if isinstance(d, torch.Tensor) else d, self.orig_shape, self._u8, self.strategy)
        m._polys, m._rects = self._polys, self._rects
        return m

    def __len__(self):
        return int(self.shape[0])

    def __getitem__(self, i):
        u = self._u8[i] if self._u8 is not None else None
        if u is not None and u.dim() == 2:
            u = u[None]
        d = None
        if self._data is not None:
            d = self._data[i]
            d = d[None] if d.ndim == 2 else d
        return Masks(d, self.orig_shape, u, self.strategy)

    def _contour(self, i: int) -> np.ndarray:
        """polygon of mask i (float32 [m,2], original-image pixels), computed on first use: the reference's loop touches ONE mask per
        frame (`masks.xy[best]`, yolo_seg/app.py:101), so nothing is traced for the others."""
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        if i in self._polys:
            return self._polys[i]
        mh, mw = self.shape[1:]
        poly, rect = None, None
        u8 = self._u8
        one = None
        if u8 is None and isinstance(self._data, torch.Tensor) and self._data.is_cuda:
            one = (self._data[i:i + 1] > 0.5).to(torch.uint8)
        elif u8 is not None and u8.is_cuda:
            one = u8[i:i + 1]
        if one is not None:
            if self.strategy == "all_merged":             # the device lists every external contour ("all") with its length; bridged on the host
                polys, rects, parts = mask_contours_device(one, strategy="all", want_parts=True)
                poly, rect = polys[0], rects[0]
                if poly is not None and len(parts[0]) > 1:
                    cuts = np.cumsum(parts[0])[:-1]
                    poly = hostops.merge_contours(np.split(poly, cuts)).astype(np.int32)
            else:
                polys, rects = mask_contours_device(one, strategy=self.strategy)
                poly, rect = polys[0], rects[0]
        if poly is None:                                  # (no GPU copy, or the device pass declined this mask: host trace)
            d = self.data[i]
            host = d.detach().cpu().numpy() if isinstance(d, torch.Tensor) else np.asarray(d)
            poly = hostops.mask_polygon(host > 0.5, self.strategy)
            rect = None
        if rect is not None and (mh, mw) == self.orig_shape:
            self._rects[i] = (float(rect[0]), float(rect[1]))   # (the device rectangle is of the polygon in mask pixels)
        if poly.shape[0] and (mh, mw) != self.orig_shape:
            poly = hostops.scale_coords((mh, mw), poly, self.orig_shape)
        self._polys[i] = poly.astype(np.float32)
        return self._polys[i]

    @property
    def xy(self) -> "_LazyPolygons":
        return _LazyPolygons(self)

    def min_rect_len(self, i: int):
        """(length, length / width) exactly as `get_coord_min_rect_len(self.xy[i])` returns them (yolo_seg/utils/mask_tools.py:12-22),
        from the device rectangle when there is one."""
        poly = self._contour(i)
        if i < 0:
            i += len(self)
        if i not in self._rects or len(poly) < 3:
            return hostops.get_coord_min_rect_len(poly)
        length, width = self._rects[i]
        if width == 0:
            width = 1
        return length, length / width


class _LazyPolygons:
    """What `Masks.xy` returns: behaves like the list of polygons ultralytics builds (len, indexing, slicing, iteration), but a
    polygon is traced only when it is asked for."""

    def __init__(self, masks: Masks):
        self._m = masks

    def __len__(self):
        return len(self._m)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._m._contour(j) for j in range(*i.indices(len(self._m)))]
        return self._m._contour(int(i))

    def __iter__(self):
        return (self._m._contour(j) for j in range(len(self._m)))


class Results:
    def __init__(self, orig_img: np.ndarray, boxes: Boxes, masks: Optional[Masks], names: Dict[int, str], path: str = ""):
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_img.shape[:2])
        self.boxes = boxes
        self.masks = masks
        self.names = names
        self.path = path

    def __len__(self):
        return len(self.boxes)

    def cpu(self):
        return Results(self.orig_img, self.boxes.cpu(), self.masks.cpu() if self.masks is not None else None, self.names, self.path)

    def numpy(self):
        return Results(self.orig_img, self.boxes.numpy(), self.masks.numpy() if self.masks is not None else None, self.names, self.path)


class _ModelHandle:
    """What callers touch through `yolo.model`: `.parameters()` for the device and an idempotent `.to()`
    (reference yolo_seg/yolo_with_deva.py:42,129-130 moves the model every frame)."""

    def __init__(self, owner: "YOLO"):
        self._owner = owner
        self.names = owner.names

    def parameters(self):
        yield self._owner._device_token()

    def to(self, device=None, *a, **k):
        self._owner._set_device(device)
        return self

    def eval(self):
        return self

    def fuse(self, *a, **k):
        return self


class YOLO:
    """Drop-in for `ultralytics.YOLO` on the predict path (reference yolo_seg/app.py:45-50,91; yolo_seg/yolo_with_deva.py:51,226) of YOLOv10
    detect / v10-seg checkpoints and of the YOLOv8-seg / YOLO11-seg checkpoints the reference's UI offers (yolo_seg/app.py:218-223); family,
    variant, class count and task are read from the checkpoint's state dict.

    model: path to an ultralytics-layout `.pt` (read without ultralytics, weights.py), or "synthetic:<n|s|m|b|l|x>[-seg]"
    (seeded weights, for tests/benchmarks: no checkpoint exists offline)."""

    def __init__(self, model: Union[str, os.PathLike] = "yolov10s.pt", task: Optional[str] = None, dtype: str = "bf16",
                 device: Optional[Union[int, str, torch.device]] = None, nc: int = 80, seed: int = 0):
        self.ckpt_path = str(model)
        self.dtype = dtype
        self.family = "v10"
        if self.ckpt_path.startswith("synthetic:"):
            spec = self.ckpt_path.split(":", 1)[1]              # "s", "s-seg" (YOLOv10) ; "v8n-seg", "11x-seg" (the app's families)
            base = spec.split("-")[0]
            self.seg = spec.endswith("-seg")
            self.nc = nc
            if base[:2] in ("v8", "11"):
                from .weights import synthetic_state_family
                self.family, self.variant = base[:2], base[2:]
                if not self.seg:
                    raise ValueError("the v8 / 11 families are built as segmentation models")
                self._state = synthetic_state_family(self.family, self.variant, nc, seed=seed)
            else:
                self.variant = base
                self._state = synthetic_state(self.variant, nc, self.seg, seed=seed)
            self.names = {i: str(i) for i in range(nc)}
        else:
            if not os.path.isfile(self.ckpt_path):
                raise FileNotFoundError(f"{self.ckpt_path}: no such checkpoint (nothing is downloaded)")
            st, meta = read_ultralytics_pt(self.ckpt_path)
            if meta.get("family") is None or meta.get("variant") is None or meta.get("nc") is None:
                raise ValueError(f"{self.ckpt_path}: not a checkpoint this engine understands (YOLOv10 detect / v10-seg, "
                                 "YOLOv8-seg, YOLO11-seg)")
            if meta["family"] != "v10" and not meta["seg"]:
                raise ValueError(f"{self.ckpt_path}: YOLOv8 / YOLO11 detect-only checkpoints are not built; the reference uses the -seg models")
            self.family = meta["family"]
            self.variant, self.seg, self.nc = meta["variant"], bool(meta["seg"]), int(meta["nc"])
            self._state = st
            names = meta.get("names")
            self.names = dict(names) if isinstance(names, dict) else {i: str(i) for i in range(self.nc)}
        if task == "segment" and not self.seg:
            raise ValueError("task='segment' but the checkpoint has no Proto/cv4 head")
        self.task = "segment" if self.seg else "detect"
        self._dev_index: Optional[int] = None
        self._set_device(device)
        self.model = _ModelHandle(self)

    # -- device / engine -----------------------------------------------------------------------------------------
    def _set_device(self, device) -> None:
        if device is None or device == "":
            idx = self._dev_index if self._dev_index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        else:
            d = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            if d.type != "cuda":
                raise ValueError(f"device={device!r}: this engine runs on MI355X GPUs only (device='cuda' / 'cuda:N' / N)")
            idx = d.index if d.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self._dev_index = int(idx)

    def _device_token(self) -> torch.Tensor:
        return torch.empty(0, device=torch.device("cuda", self._dev_index))

    def _engine(self) -> Engine:
        # the reference rebuilds the model on every request (yolo_seg/app.py:45): keep engines per (ckpt, mtime, ...)
        try:
            mt = os.path.getmtime(self.ckpt_path)
        except OSError:
            mt = 0.0
        key = (self.ckpt_path, mt, self.family, self.variant, self.nc, self.seg, self.dtype, self._dev_index)
        eng = _ENGINE_CACHE.get(key)
        if eng is None:
            eng = Engine(self.variant, self.nc, self.seg, self.dtype, self._dev_index, state=self._state, family=self.family)
            _ENGINE_CACHE[key] = eng
        return eng

    # -- predict ---------------------------------------------------------------------------------------------------
    def __call__(self, source=None, **kw):
        return self.predict(source, **kw)

    def predict(self, source=None, conf: float = 0.25, retina_masks: bool = False, device=None, imgsz: int = 640,
                max_det: int = 300, stream: bool = False, verbose: bool = False, iou: float = 0.7, **ignored) -> List[Results]:
        if source is None:
            raise ValueError("predict() needs a source (ndarray BGR HWC uint8, PIL.Image, path or a list of them)")
        if device is not None:
            self._set_device(device)
        imgs, paths = hostops.load_sources(source)                    # list of BGR uint8 HWC (ndarray => taken as BGR)
        eng = self._engine()
        if self.family != "v10":
            eng.set_nms(conf, iou)                                    # v8 / 11: conf filter + NMS run inside the forward
        dev = torch.device("cuda", self._dev_index)
        results: List[Results] = []
        # frames that letterbox to the same shape run as one batch
        groups: Dict[Tuple[int, int], List[int]] = {}
        geos = []
        for i, im in enumerate(imgs):
            geo = hostops.letterbox_geometry(im.shape[0], im.shape[1], imgsz)
            geos.append(geo)
            groups.setdefault((geo["out_h"], geo["out_w"]), []).append(i)
        out_by_index: Dict[int, Results] = {}
        for (H, W), idxs in groups.items():
            # the raw frame goes up once; resize + pad-114 run on the device straight into the batch tensor (yp_letterbox)
            # (one persistent batch buffer per shape: the engine's hipGraph is specialised on the input pointer)
            bcache = self.__dict__.setdefault("_batch_cache", {})
            bkey = (self._dev_index, len(idxs), H, W)
            batch = bcache.get(bkey)
            if batch is None:
                batch = bcache[bkey] = torch.empty((len(idxs), H, W, 3), dtype=torch.uint8, device=dev)
            for bi, i in enumerate(idxs):
                raw = torch.from_numpy(np.ascontiguousarray(imgs[i])).to(dev, non_blocking=True)
                letterbox_device(raw, geos[i], out=batch[bi])
            # one output set per batch size (no allocation per call); everything handed to the caller below is copied out of it
            cache = self.__dict__.setdefault("_out_cache", {})
            okey = (self._dev_index, len(idxs))
            # launch mode "auto": per input shape the engine times eager launches against hipGraph replay once and keeps the faster - the
            # reference's one-frame calls (yolo_seg/app.py:85-91) run eagerly (~80 kernels of a few microseconds: the graph executor's
            # per-node cost loses), batches replay. YOLOP_PREDICT_GRAPH=0 / 1 forces one mode.
            mode = os.environ.get("YOLOP_PREDICT_GRAPH", "auto")
            eng.set_graph("auto" if mode == "auto" else int(mode))
            out = eng.forward(batch, cache.get(okey))
            cache[okey] = out
            # ONE device-to-host copy of the [B,300,6] rows ends the forward; the conf filter and scale_boxes are a few dozen floats of
            # host arithmetic (a chain of tiny device ops with two boolean gathers - each a device sync - cost ~0.1 ms per frame).
            # Both heads emit their rows best first, so the rows above `conf` are a prefix and the coefficients are sliced, not gathered.
            det_host = out["det"][:len(idxs)].cpu()
            for bi, i in enumerate(idxs):
                oh, ow = imgs[i].shape[:2]
                dh = det_host[bi]
                keep = dh[:, 4] > conf                                  # strict, A.6 step 2
                n = int(keep.sum())
                prefix = bool(keep[:n].all())
                d = (dh[:n] if prefix else dh[keep])[:max_det].clone()
                n = int(d.shape[0])
                boxes_in = d[:, :4].clone()                              # letterboxed-input pixels
                d[:, :4] = hostops.scale_boxes_t((H, W), d[:, :4], (oh, ow))
                masks = None
                if self.seg and n > 0:
                    cf = out["coeff"][bi][:n] if prefix else out["coeff"][bi][keep.to(dev)][:max_det]
                    if retina_masks:
                        m, _, _ = eng.masks(bi, cf, d[:, :4].to(dev, non_blocking=True), (oh, ow), retina=True)
                    else:
                        m, _, _ = eng.masks(bi, cf, boxes_in.to(dev, non_blocking=True), (H, W), retina=False)
                    masks = Masks(None, (oh, ow), u8=m)
                out_by_index[i] = Results(imgs[i], Boxes(d, (oh, ow)), masks, self.names, paths[i])
        for i in range(len(imgs)):
            results.append(out_by_index[i])
        return results

    def predict_id_mask(self, image: np.ndarray, conf: float = 0.9, out_hw: Optional[Tuple[int, int]] = None,
                        suppress_small: bool = False, min_area: int = 100, imgsz: int = 640,
                        pre_resize: Optional[Tuple[int, int]] = None):
        """The fused body of `auto_segment` (reference yolo_seg/yolo_with_deva.py:45-86): (cv2.resize to `pre_resize` = (w,h), :45-48)
        -> predict(retina_masks=True, conf) -> masks at the fed frame's size -> (antialiased bilinear to `out_hw` if it differs,
        :71-72) -> area test -> id paint, all on the GPU: the raw frame is uploaded once.
        -> (ids int64 [h,w] cuda, kept int32 [n] cpu (id per detection, 0 = suppressed), conf [n] cpu, cls [n] cpu)"""
        if not self.seg:
            raise ValueError("auto_segment needs a segmentation checkpoint")
        imgs, _ = hostops.load_sources(image)
        im = imgs[0]
        eng = self._engine()
        dev = torch.device("cuda", self._dev_index)
        raw = torch.from_numpy(np.ascontiguousarray(im)).to(dev)
        if pre_resize is not None and (int(pre_resize[1]), int(pre_resize[0])) != tuple(im.shape[:2]):
            nw, nh = int(pre_resize[0]), int(pre_resize[1])
            raw = letterbox_device(raw, dict(out_h=nh, out_w=nw, new_h=nh, new_w=nw, top=0, left=0))   # cv2.resize(INTER_LINEAR) on the device
        oh, ow = int(raw.shape[0]), int(raw.shape[1])
        out_hw = (oh, ow) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
        geo = hostops.letterbox_geometry(oh, ow, imgsz)
        H, W = geo["out_h"], geo["out_w"]
        bcache = self.__dict__.setdefault("_batch_cache", {})
        batch = bcache.get((self._dev_index, 1, H, W))
        if batch is None:
            batch = bcache[(self._dev_index, 1, H, W)] = torch.empty((1, H, W, 3), dtype=torch.uint8, device=dev)
        letterbox_device(raw, geo, out=batch[0])
        if self.family != "v10":
            eng.set_nms(conf, 0.7)
        mode = os.environ.get("YOLOP_PREDICT_GRAPH", "auto")             # (explicit: not whatever the last predict() left; one frame resolves to eager)
        eng.set_graph("auto" if mode == "auto" else int(mode))
        out = eng.forward(batch)
        dh = out["det"][0].cpu()                                        # (one copy; see predict())
        keep = dh[:, 4] > conf
        n = int(keep.sum())
        prefix = bool(keep[:n].all())
        d = dh[:n] if prefix else dh[keep]
        if n == 0:
            z = torch.zeros(out_hw, dtype=torch.int64, device=dev)
            e = torch.zeros(0)
            return z, torch.zeros(0, dtype=torch.int32), e, e
        boxes = hostops.scale_boxes_t((H, W), d[:, :4], (oh, ow)).to(dev, non_blocking=True)
        cf = out["coeff"][0][:n] if prefix else out["coeff"][0][keep.to(dev)]
        if out_hw == (oh, ow):
            _, ids, kept = eng.masks(0, cf, boxes, (oh, ow), retina=True, want_masks=False, want_ids=True,
                                     suppress_small=suppress_small, min_area=min_area)
        else:
            # the reference resizes each float mask to (h,w) (torchvision F.resize: antialiased bilinear), tests the float area and
            # thresholds at 0.5 (:71-79): one GPU pass (yp_id_mask_resized)
            ids, kept = eng.id_mask_resized(0, cf, boxes, (oh, ow), out_hw, suppress_small=suppress_small, min_area=min_area)
        return ids, kept.cpu(), d[:, 4].clone(), d[:, 5].clone()
