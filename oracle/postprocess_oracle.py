"""CPU ORACLE (test infrastructure, NOT product code): everything of the predict path that sits around the
network - letterbox geometry, box rescale, the segmentation mask tail, and the one in-tree function on the path,
`auto_segment` (reference yolo_seg/yolo_with_deva.py:37-88, restated from its source text).

PARITY UNPINNED for the ultralytics-side functions (see yolov10_oracle.py header): they restate SURVEY.md
Appendix A.5-A.7 [U]. `auto_segment_oracle` and `select_best_box` follow in-tree reference code and cite it
line by line, but the reference has no fixtures for them either.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ---- A.5 LetterBox geometry -----------------------------------------------------------------------------------
def letterbox_geometry(h0: int, w0: int, new_shape: int = 640, stride: int = 32, auto: bool = True,
                       scaleup: bool = True) -> dict:
    """[U] LetterBox(new_shape, auto, stride, center=True): returns resize size and integer pads.
    r=min(640/h,640/w); new_unpad=(round(w*r),round(h*r)); dw,dh = 640-new_unpad; auto: mod stride; halve;
    top=round(dh-0.1) bottom=round(dh+0.1) left=round(dw-0.1) right=round(dw+0.1)."""
    r = min(new_shape / h0, new_shape / w0)
    if not scaleup:
        r = min(r, 1.0)
    nw, nh = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = new_shape - nw, new_shape - nh
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return dict(r=r, new_w=nw, new_h=nh, top=top, bottom=bottom, left=left, right=right,
                out_h=nh + top + bottom, out_w=nw + left + right)


def resize_bilinear_u8_cv2(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """Restatement of OpenCV's 8-bit INTER_LINEAR resize (fixed-point: 11-bit coefficients, the
    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2 vertical pass). cv2 is not installed here,
    so this is recalled from the public OpenCV source and NOT verified against cv2: UNPINNED."""
    h0, w0, cn = img.shape
    if (h0, w0) == (new_h, new_w):
        return img.copy()
    COEF = 2048

    def axis(n_dst, n_src):
        scale = n_src / n_dst
        d = np.arange(n_dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s).astype(np.float32)
        lo = s < 0
        s[lo] = 0
        f[lo] = 0
        hi = s >= n_src - 1
        s[hi] = n_src - 1
        f[hi] = 0
        a1 = np.rint(f * COEF).astype(np.int64)          # saturate_cast<short>(fx*2048)
        a0 = np.rint((1.0 - f) * COEF).astype(np.int64)
        s1 = np.minimum(s + 1, n_src - 1)
        return s, s1, a0, a1

    sx0, sx1, ax0, ax1 = axis(new_w, w0)
    sy0, sy1, ay0, ay1 = axis(new_h, h0)
    src = img.astype(np.int64)
    # horizontal pass (int32 rows, values scaled by 2048)
    hrow = src[:, sx0, :] * ax0[None, :, None] + src[:, sx1, :] * ax1[None, :, None]
    r0 = hrow[sy0]
    r1 = hrow[sy1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img_bgr: np.ndarray, new_shape: int = 640, stride: int = 32, auto: bool = True) -> Tuple[np.ndarray, dict]:
    """uint8 HxWx3 -> letterboxed uint8 (pad value 114) + geometry."""
    g = letterbox_geometry(img_bgr.shape[0], img_bgr.shape[1], new_shape, stride, auto)
    im = resize_bilinear_u8_cv2(img_bgr, g["new_w"], g["new_h"])
    out = np.full((g["out_h"], g["out_w"], 3), 114, dtype=np.uint8)
    out[g["top"]:g["top"] + g["new_h"], g["left"]:g["left"] + g["new_w"]] = im
    return out, g


# ---- A.6 steps 2-4 ---------------------------------------------------------------------------------------------
def scale_boxes(img1_hw: Sequence[int], boxes: Tensor, img0_hw: Sequence[int]) -> Tensor:
    """[U] ops.scale_boxes + clip_boxes: undo letterbox pad/gain, clamp to the original image."""
    gain = min(img1_hw[0] / img0_hw[0], img1_hw[1] / img0_hw[1])
    padx = round((img1_hw[1] - img0_hw[1] * gain) / 2 - 0.1)
    pady = round((img1_hw[0] - img0_hw[0] * gain) / 2 - 0.1)
    b = boxes.clone()
    b[..., 0] -= padx
    b[..., 2] -= padx
    b[..., 1] -= pady
    b[..., 3] -= pady
    b[..., :4] /= gain
    b[..., 0].clamp_(0, img0_hw[1])
    b[..., 2].clamp_(0, img0_hw[1])
    b[..., 1].clamp_(0, img0_hw[0])
    b[..., 3].clamp_(0, img0_hw[0])
    return b


def conf_filter(det: Tensor, conf: float) -> Tensor:
    """[U] A.6 step 2: keep = score > conf (strict). det [k,6] -> [n,6] (order preserved = descending score)."""
    return det[det[:, 4] > conf]


def xyxy2xywhn(xyxy: Tensor, orig_hw: Sequence[int]) -> Tensor:
    """[U] Boxes.xywhn (reference dev_tools/classify/cls_bbox_dataset_generate.py:52)."""
    x1, y1, x2, y2 = xyxy.unbind(-1)
    out = torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), -1)
    return out / torch.tensor([orig_hw[1], orig_hw[0], orig_hw[1], orig_hw[0]], dtype=out.dtype)


# ---- A.7 segmentation tail ---------------------------------------------------------------------------------------
def crop_mask(masks: Tensor, boxes: Tensor) -> Tensor:
    """[U] zero where not (x>=x1)&(x<x2)&(y>=y1)&(y<y2), float box edges."""
    _, h, w = masks.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, dtype=x1.dtype)[None, :, None]
    return masks * ((r >= x1) * (r < x2) * (c >= y1) * (c < y2))


def scale_masks_region(mh: int, mw: int, oh: int, ow: int) -> Tuple[int, int, int, int]:
    """[U] scale_masks crop rectangle in proto pixels: (top, bottom, left, right)."""
    gain = min(mh / oh, mw / ow)
    padw = (mw - ow * gain) / 2
    padh = (mh - oh * gain) / 2
    return int(padh), int(mh - padh), int(padw), int(mw - padw)


def process_mask_native(proto: Tensor, coeff: Tensor, boxes_orig: Tensor, orig_hw: Sequence[int]) -> Tensor:
    """[U] retina_masks=True path. proto [32,mh,mw], coeff [n,32], boxes in ORIGINAL pixels -> float {0,1} [n,oh,ow]."""
    c, mh, mw = proto.shape
    m = (coeff @ proto.float().view(c, -1)).view(-1, mh, mw)
    t, b, l, r = scale_masks_region(mh, mw, orig_hw[0], orig_hw[1])
    m = F.interpolate(m[None, :, t:b, l:r], size=tuple(orig_hw), mode="bilinear", align_corners=False)[0]
    m = crop_mask(m, boxes_orig)
    return (m > 0.0).to(torch.float32)


def process_mask(proto: Tensor, coeff: Tensor, boxes_in: Tensor, in_hw: Sequence[int]) -> Tensor:
    """[U] retina_masks=False path (reference dev_tools/auto_speed_calc.py:62): crop at proto resolution,
    bilinear up to the LETTERBOXED input size, > 0."""
    c, mh, mw = proto.shape
    ih, iw = in_hw
    m = (coeff @ proto.float().view(c, -1)).view(-1, mh, mw)
    ds = boxes_in.clone()
    ds[:, 0] *= mw / iw
    ds[:, 2] *= mw / iw
    ds[:, 1] *= mh / ih
    ds[:, 3] *= mh / ih
    m = crop_mask(m, ds)
    m = F.interpolate(m[None], size=(ih, iw), mode="bilinear", align_corners=False)[0]
    return (m > 0.0).to(torch.float32)


# ---- in-tree reference functions ---------------------------------------------------------------------------------
def auto_segment_oracle(masks: Optional[Tensor], conf: Tensor, cls: Tensor, out_hw: Tuple[int, int],
                        suppress_small_mask: bool, min_area: int = 100):
    """Restates reference yolo_seg/yolo_with_deva.py:54-86 given the predict() outputs.

    :54      output_mask = zeros((h,w), int64)
    :61-62   if masks is not None: for i in range(len(masks))
    :71-72   resize to (h,w) if shapes differ (torchvision F.resize on a float mask -> bilinear)
    :75      skip if suppress_small_mask and mask.sum() < MIN_AREA_THRESHOLD(100)
    :79      output_mask[mask > 0.5] = curr_id   (later ids overwrite earlier)
    :82-85   ObjectInfo(id=curr_id, score=conf[i], category_id=int(cls[i])) ; ids consecutive over KEPT masks
    returns (int64 [h,w], list of (id, score, category_id))."""
    h, w = out_hw
    out = torch.zeros((h, w), dtype=torch.int64)
    info: List[Tuple[int, float, int]] = []
    cur = 1
    if masks is not None:
        for i in range(len(masks)):
            m = masks[i].float()
            if tuple(m.shape) != (h, w):
                m = F.interpolate(m[None, None], size=(h, w), mode="bilinear", align_corners=False,
                                  antialias=True)[0, 0]
            if suppress_small_mask and m.sum() < min_area:
                continue
            out[m > 0.5] = cur
            info.append((cur, float(conf[i]), int(cls[i])))
            cur += 1
    return out, info


def select_best_box(xyxy: np.ndarray, conf: np.ndarray, last_box, width: int, height: int):
    """Restates reference yolo_seg/app.py:95-112: arg-max-confidence box -> int xyxy; fallback to the previous
    box, or the full frame, when there is no detection. returns (box, best_index|None)."""
    if len(conf) > 0:
        best = int(np.argmax(conf))
        return list(map(int, xyxy[best].squeeze())), best
    if last_box is None:
        return (0, 0, width, height), None
    return last_box, None



# ---- A.7 masks2segments, later 8.3.x form [U] ------------------------------------------------------------------------------------
# Restated from the published ultralytics sources as recalled (the package is not in /root/reference and not installed: unpinned).
# ultralytics/utils/ops.py masks2segments(masks, strategy="all"): per mask `c = cv2.findContours(x, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0]`;
#   "largest": the contour with the most points;   "all", releases up to early 8.3: `np.concatenate([x.reshape(-1, 2) for x in c])`;
#   "all", later 8.3.x: `np.concatenate(merge_multi_segment([x.reshape(-1, 2) for x in c])) if len(c) > 1 else c[0].reshape(-1, 2)` -
#   the contours are CONNECTED at their closest points instead of being laid end to end.
# ultralytics/data/converter.py merge_multi_segment / min_index (the COCO converter's routine) is restated below line by line.
# Every point of every contour appears in the merged polygon (some twice: a closing point per contour, the bridge points), so the point
# SET, its convex hull and cv2.minAreaRect - what the reference reads out of `masks.xy[best]` (yolo_seg/app.py:101-103 ->
# utils/mask_tools.py:12-22) - are the same for both forms; only the polygon's vertex order differs.
def min_index(arr1: np.ndarray, arr2: np.ndarray) -> Tuple[int, int]:
    dis = ((arr1[:, None, :] - arr2[None, :, :]) ** 2).sum(-1)
    i = int(np.argmin(dis, axis=None))                      # first minimum in row-major order
    return i // dis.shape[1], i % dis.shape[1]


def merge_multi_segment(segments: Sequence[np.ndarray]) -> list:
    s = []
    segments = [np.array(i).reshape(-1, 2) for i in segments]
    idx_list = [[] for _ in range(len(segments))]
    for i in range(1, len(segments)):                       # closest pair of points between consecutive contours
        idx1, idx2 = min_index(segments[i - 1], segments[i])
        idx_list[i - 1].append(idx1)
        idx_list[i].append(idx2)
    for k in range(2):                                      # forward pass, then the way back over the middle contours
        if k == 0:
            for i, idx in enumerate(idx_list):
                if len(idx) == 2 and idx[0] > idx[1]:       # a middle contour is walked from its entry point to its exit point
                    idx = idx[::-1]
                    segments[i] = segments[i][::-1, :]
                segments[i] = np.roll(segments[i], -idx[0], axis=0)
                segments[i] = np.concatenate([segments[i], segments[i][:1]])
                if i in {0, len(idx_list) - 1}:             # first and last contour: whole, closed
                    s.append(segments[i])
                else:
                    idx = [0, idx[1] - idx[0]]
                    s.append(segments[i][idx[0]: idx[1] + 1])
        else:
            for i in range(len(idx_list) - 1, -1, -1):
                if i not in {0, len(idx_list) - 1}:
                    idx = idx_list[i]
                    nidx = abs(idx[1] - idx[0])
                    s.append(segments[i][nidx:])
    return s


def masks2segments_contours(contours: Sequence[np.ndarray], strategy: str = "all", merged: bool = True) -> np.ndarray:
    """masks2segments [U] applied to the contour list cv2.findContours would return for one mask (list order = cv2's: last found first).
    merged=True: the later 8.3.x "all" (merge_multi_segment); merged=False: plain concatenation (earlier releases)."""
    c = [np.asarray(x).reshape(-1, 2) for x in contours]
    if not c:
        return np.zeros((0, 2), dtype=np.float32)
    if strategy == "all":
        out = (np.concatenate(merge_multi_segment(c)) if merged else np.concatenate(c)) if len(c) > 1 else c[0]
    elif strategy == "largest":
        out = c[int(np.array([len(x) for x in c]).argmax())]
    else:
        raise ValueError(strategy)
    return out.astype(np.float32)
