"""Host-side steps around the engine: source loading, LetterBox, box rescale, mask -> polygon.

These sit on the CPU in the reference too (cv2 / numpy inside ultralytics' predictor; SURVEY.md A.5-A.7 [U]) and are
"next" rows of the scope table for a GPU version. cv2 is not available offline, so the 8-bit
INTER_LINEAR resize and the external-contour trace are restated here from the public OpenCV algorithms.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch


# ---- sources (A.5 step 1) --------------------------------------------------------------------------------------------
def _one_source(src) -> Tuple[np.ndarray, str]:
    if isinstance(src, np.ndarray):
        if src.ndim != 3 or src.shape[2] != 3 or src.dtype != np.uint8:
            raise TypeError(f"ndarray source must be HxWx3 uint8 (BGR), got {src.shape} {src.dtype}")
        return src, ""                                    # ndarray => taken as BGR, never auto-corrected (SURVEY C-6)
    if isinstance(src, (str, os.PathLike)):
        from PIL import Image
        p = os.fspath(src)
        if not os.path.isfile(p):
            raise FileNotFoundError(p)
        with Image.open(p) as im:
            rgb = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1]), p
    if hasattr(src, "convert") and hasattr(src, "size"):   # PIL.Image -> RGB -> BGR (yolo_seg/app.py:49)
        rgb = np.asarray(src.convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1]), ""
    if isinstance(src, torch.Tensor):
        raise TypeError("tensor sources are not part of the reference's call sites; pass ndarray / PIL / path")
    raise TypeError(f"unsupported source type {type(src)}")


def load_sources(source) -> Tuple[List[np.ndarray], List[str]]:
    items = list(source) if isinstance(source, (list, tuple)) else [source]
    if not items:
        raise ValueError("empty source list")
    pairs = [_one_source(s) for s in items]
    return [p[0] for p in pairs], [p[1] for p in pairs]


# ---- LetterBox (A.5 step 2) ------------------------------------------------------------------------------------------
def letterbox_geometry(h0: int, w0: int, new_shape: int = 640, stride: int = 32, auto: bool = True) -> dict:
    r = min(new_shape / h0, new_shape / w0)
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


def _axis_coeffs(n_dst: int, n_src: int):
    scale = n_src / n_dst
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= n_src - 1
    s[lo], f[lo] = 0, 0
    s[hi], f[hi] = n_src - 1, 0
    a1 = np.rint(f * 2048).astype(np.int64)
    a0 = np.rint((np.float32(1.0) - f) * 2048).astype(np.int64)
    return s, np.minimum(s + 1, n_src - 1), a0, a1


def resize_linear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """8-bit bilinear resize in OpenCV's fixed-point form (11-bit coefficients; vertical pass
    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2). Restated, not verified against cv2 (absent offline)."""
    h0, w0 = img.shape[:2]
    if (h0, w0) == (new_h, new_w):
        return img
    x0, x1, ax0, ax1 = _axis_coeffs(new_w, w0)
    y0, y1, ay0, ay1 = _axis_coeffs(new_h, h0)
    src = img.astype(np.int64)
    hrow = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]
    out = (((ay0[:, None, None] * (hrow[y0] >> 4)) >> 16) + ((ay1[:, None, None] * (hrow[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img_bgr: np.ndarray, new_shape: int = 640, stride: int = 32, auto: bool = True):
    g = letterbox_geometry(img_bgr.shape[0], img_bgr.shape[1], new_shape, stride, auto)
    im = resize_linear_u8(img_bgr, g["new_w"], g["new_h"])
    out = np.full((g["out_h"], g["out_w"], 3), 114, dtype=np.uint8)
    out[g["top"]:g["top"] + g["new_h"], g["left"]:g["left"] + g["new_w"]] = im
    return out, g


# ---- boxes (A.6 step 3) ----------------------------------------------------------------------------------------------
def scale_boxes_t(img1_hw: Sequence[int], boxes: torch.Tensor, img0_hw: Sequence[int]) -> torch.Tensor:
    """undo letterbox pad/gain, clamp to the original image (ops.scale_boxes + clip_boxes)."""
    gain = min(img1_hw[0] / img0_hw[0], img1_hw[1] / img0_hw[1])
    padx = round((img1_hw[1] - img0_hw[1] * gain) / 2 - 0.1)
    pady = round((img1_hw[0] - img0_hw[0] * gain) / 2 - 0.1)
    b = boxes.clone()
    b[:, 0] -= padx
    b[:, 2] -= padx
    b[:, 1] -= pady
    b[:, 3] -= pady
    b /= gain
    b[:, 0].clamp_(0, img0_hw[1])
    b[:, 2].clamp_(0, img0_hw[1])
    b[:, 1].clamp_(0, img0_hw[0])
    b[:, 3].clamp_(0, img0_hw[0])
    return b


def scale_coords(img1_hw: Sequence[int], coords: np.ndarray, img0_hw: Sequence[int]) -> np.ndarray:
    """polygon points from the letterboxed frame to the original image (ops.scale_coords)."""
    gain = min(img1_hw[0] / img0_hw[0], img1_hw[1] / img0_hw[1])
    padx = (img1_hw[1] - img0_hw[1] * gain) / 2
    pady = (img1_hw[0] - img0_hw[0] * gain) / 2
    c = coords.astype(np.float32).copy()
    c[:, 0] = np.clip((c[:, 0] - padx) / gain, 0, img0_hw[1])
    c[:, 1] = np.clip((c[:, 1] - pady) / gain, 0, img0_hw[0])
    return c


# ---- Masks.xy (A.7): cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + masks2segments, restated ----------------------------
# Definitions this module and the device kernel (csrc/contour.hip) both implement - cv2 is absent here, so they are stated, not pinned:
#   * foreground 8-connected, background 4-connected (the pairing of Suzuki & Abe's border following, which cv2.findContours implements);
#   * an OUTER border belongs to one 8-connected blob; it is EXTERNAL (what RETR_EXTERNAL keeps) iff the blob is not enclosed by another
#     blob, i.e. iff the background pixel west of the blob's raster-first pixel is 4-connected to the image frame. A blob inside a hole of
#     another blob is NOT external and is skipped, as cv2 skips it;
#   * a contour starts at its blob's raster-first pixel and runs DOWN first (counter-clockwise on screen: for a filled rectangle top-left,
#     bottom-left, bottom-right, top-right - the order cv2 returns); CHAIN_APPROX_SIMPLE keeps a pixel iff the move into it differs from
#     the move out of it (the start pixel included, judged on the closed chain);
#   * contours are listed bottom-up: descending raster order of their start pixels (cv2 returns the last contour it found first);
#   * masks2segments [U]: strategy "largest" = the contour with the most points (first in list order on a tie), "all" (the default of the
#     8.3.x line that YOLO11 weights need) = all contours concatenated in list order.
_DIRS = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]   # (dy,dx), clockwise from east


def _trace_outer(mask: np.ndarray, sy: int, sx: int) -> List[Tuple[int, int]]:
    """Moore-neighbour boundary trace (8-connectivity) of the blob containing (sy,sx), which must be its first pixel
    in raster order; Jacob's stopping criterion. Clockwise on screen (east first); `_cv_order` turns the chain round."""
    H, W = mask.shape

    def on(y, x):
        return 0 <= y < H and 0 <= x < W and mask[y, x]

    pts = [(sx, sy)]
    cy, cx, d = sy, sx, 6            # we "arrived" from the north-west side: start searching at north
    start_d = None
    for _ in range(4 * H * W + 8):
        found = False
        for k in range(8):
            nd = (d + k) % 8
            ny, nx = cy + _DIRS[nd][0], cx + _DIRS[nd][1]
            if on(ny, nx):
                if start_d is None:
                    start_d = nd
                elif (cy, cx) == (sy, sx) and nd == start_d:
                    return pts[:-1] if len(pts) > 1 else pts
                cy, cx = ny, nx
                pts.append((cx, cy))
                d = (nd + 6) % 8 if (nd % 2 == 0) else (nd + 5) % 8   # restart at the background pixel examined last
                found = True
                break
        if not found:
            return pts                  # isolated pixel
    return pts


def _compress(pts: List[Tuple[int, int]]) -> np.ndarray:
    """CHAIN_APPROX_SIMPLE on a closed chain in the order given: keep only the end points of horizontal / vertical / diagonal runs."""
    n = len(pts)
    if n <= 2:
        return np.asarray(pts, dtype=np.int32).reshape(-1, 2)
    keep = []
    for i in range(n):
        px, py = pts[i - 1]
        cx, cy = pts[i]
        nx, ny = pts[(i + 1) % n]
        if (cx - px, cy - py) != (nx - cx, ny - cy):
            keep.append((cx, cy))
    if not keep:
        keep = [pts[0]]
    return np.asarray(keep, dtype=np.int32)


def _cv_order(pts: List[Tuple[int, int]]) -> np.ndarray:
    """The clockwise chain of `_trace_outer` walked the other way round from the same start pixel (the direction cv2 follows an outer
    border in), then CHAIN_APPROX_SIMPLE. The kept pixels are the same set - a pixel is a run end in either direction."""
    if len(pts) <= 2:
        return _compress(pts)
    return _compress([pts[0]] + pts[:0:-1])


def external_contours(mask: np.ndarray) -> List[np.ndarray]:
    """cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0] as defined at the top of this section: list of int32 [m,2] (x,y)."""
    from scipy import ndimage
    mask = np.asarray(mask, dtype=bool)
    lab, n = ndimage.label(mask, structure=np.ones((3, 3), dtype=int))
    if n == 0:
        return []
    holes = ndimage.binary_fill_holes(mask) & ~mask          # (default structure: the background is invaded 4-connectedly from the frame)
    found = []
    for k, sl in enumerate(ndimage.find_objects(lab), start=1):
        sub = lab[sl] == k
        ys, xs = np.nonzero(sub)
        sy = ys.min()
        sx = xs[ys == sy].min()
        gy, gx = int(sy) + sl[0].start, int(sx) + sl[1].start
        if gx > 0 and holes[gy, gx - 1]:
            continue                                         # the blob sits in a hole of another blob: not an external contour
        poly = _cv_order(_trace_outer(sub, int(sy), int(sx))) + np.asarray([sl[1].start, sl[0].start], dtype=np.int32)
        found.append((gy * mask.shape[1] + gx, poly))
    found.sort(key=lambda t: -t[0])
    return [p for _, p in found]


POLYGON_STRATEGIES = ("all", "largest", "all_merged")


def merge_contours(contours: List[np.ndarray]) -> np.ndarray:
    """"all" of the LATER 8.3.x releases [U]: `masks2segments` there joins several contours with `merge_multi_segment` (the COCO converter's
    routine) instead of laying them end to end - consecutive contours of the list are bridged at their closest pair of points (first
    minimum in row-major order of the squared-distance table), the first and the last contour are walked whole and closed, a middle one
    from its entry point to its exit point on the way out and the rest of it on the way back. Same point set as the concatenation (hence
    the same hull and minimum-area rectangle), another vertex order."""
    segs = [np.asarray(c).reshape(-1, 2) for c in contours]
    n = len(segs)
    if n == 0:
        return np.zeros((0, 2), dtype=np.int32)
    if n == 1:
        return segs[0]
    near = [[] for _ in range(n)]
    for i in range(1, n):
        a, b = segs[i - 1].astype(np.int64), segs[i].astype(np.int64)
        d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        j = int(d.argmin())
        near[i - 1].append(j // d.shape[1])
        near[i].append(j % d.shape[1])
    out, back = [], []
    for i in range(n):
        idx = near[i]
        seg = segs[i]
        if len(idx) == 2 and idx[0] > idx[1]:
            idx, seg = idx[::-1], seg[::-1]
        seg = np.roll(seg, -idx[0], axis=0)
        seg = np.concatenate([seg, seg[:1]])
        if i == 0 or i == n - 1:
            out.append(seg)
        else:
            out.append(seg[:idx[1] - idx[0] + 1])
            back.append(seg[abs(near[i][1] - near[i][0]):])
    return np.concatenate(out + back[::-1], axis=0)


def mask_polygon(mask: np.ndarray, strategy: str = "all") -> np.ndarray:
    """ultralytics `masks2segments(mask, strategy)` for one mask [U]: int32 [m,2] (x,y); [0,2] for an empty mask. "all" = every external
    contour laid end to end, "all_merged" = the same contours bridged at their closest points (later 8.3.x, merge_contours)."""
    if strategy not in POLYGON_STRATEGIES:
        raise ValueError(f"strategy must be one of {POLYGON_STRATEGIES}, got {strategy!r}")
    c = external_contours(mask)
    if not c:
        return np.zeros((0, 2), dtype=np.int32)
    if strategy == "all":
        return np.concatenate(c, axis=0)
    if strategy == "all_merged":
        return merge_contours(c).astype(np.int32)
    return c[int(np.argmax([len(x) for x in c]))]


def largest_external_contour(mask: np.ndarray) -> np.ndarray:
    """[m,2] (x,y) int32 polygon: the external contour with the most points (ultralytics masks2segments 'largest')."""
    return mask_polygon(mask, "largest")


# ---- shaft length from a polygon (reference yolo_seg/utils/mask_tools.py:12-22, which calls cv2.minAreaRect) ---------------------
def _convex_hull(pts: np.ndarray) -> np.ndarray:
    """Andrew's monotone chain on integer points; counter-clockwise, no collinear points, float64 out."""
    p = np.unique(pts.astype(np.int64), axis=0)            # sorted by x, then y
    if p.shape[0] <= 2:
        return p.astype(np.float64)

    def half(seq):
        h = []
        for q in seq:
            while len(h) >= 2 and (h[-1][0] - h[-2][0]) * (q[1] - h[-2][1]) - (h[-1][1] - h[-2][1]) * (q[0] - h[-2][0]) <= 0:
                h.pop()
            h.append(q)
        return h

    lo, up = half(list(map(tuple, p))), half(list(map(tuple, p[::-1])))
    return np.asarray(lo[:-1] + up[:-1], dtype=np.float64)


def min_area_rect_size(points) -> Tuple[float, float]:
    """(long side, short side) of the minimum-area enclosing rectangle of integer points: one side of that rectangle is collinear
    with an edge of the convex hull (rotating calipers; the public algorithm behind cv2.minAreaRect, here in float64)."""
    hull = _convex_hull(np.asarray(points).reshape(-1, 2))
    n = hull.shape[0]
    if n == 0 or n == 1:
        return 0.0, 0.0
    if n == 2:
        return float(np.hypot(*(hull[1] - hull[0]))), 0.0
    best = None
    for i in range(n):
        e = hull[(i + 1) % n] - hull[i]
        u = e / np.hypot(e[0], e[1])
        v = np.array([-u[1], u[0]])
        a, b = hull @ u, hull @ v
        w, h = a.max() - a.min(), b.max() - b.min()
        if best is None or w * h < best[0]:
            best = (w * h, w, h)
    return float(max(best[1], best[2])), float(min(best[1], best[2]))


def get_coord_min_rect_len(coord_xy) -> Tuple[float, float]:
    """Drop-in for the reference's `get_coord_min_rect_len(results[0].masks.xy[i])` (yolo_seg/app.py:101-103): the polygon is
    truncated to int32 like there; -> (length of the longer side in pixels, length / shorter side, with a zero width counted as 1);
    (0, 0) for fewer than 3 points."""
    points = np.array(coord_xy, dtype=np.int32).reshape((-1, 2))
    if len(points) < 3:
        return 0, 0
    length, width = min_area_rect_size(points)
    if width == 0:
        width = 1
    return length, length / width
