"""Test infrastructure: border following as published by Suzuki & Abe (CVGIP 30, 1985, Algorithm 1) in the form OpenCV's
`findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)` runs it - raster scan, outer borders only, pixel labels +NBD / -NBD, the
"last labelled pixel met on this row is positive => we are inside a traced border => skip" rule of the external mode, chain
compression by direction change. Written from the paper and from memory of OpenCV's scanner (cv2 is not installed: unpinned);
it is a THIRD statement of the semantics next to hostops.external_contours (connected components + hole filling) and the
device kernel (parallel Moore segments + crossing parity), and shares no code with either. Pure-Python loops: small masks only."""
from typing import List

import numpy as np

# direction codes 0..7 = E, NE, N, NW, W, SW, S, SE (counter-clockwise on screen), as (dx, dy)
_D = [(1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1)]


def _fetch_outer(img: np.ndarray, x0: int, y0: int, nbd: int) -> List[tuple]:
    """Follow the outer border that starts at (x0, y0) (its west neighbour is 0), label it, return the CHAIN_APPROX_SIMPLE points."""
    s_end = s = 4
    while True:                                  # first non-zero neighbour, clockwise from north-west
        s = (s - 1) & 7
        x1, y1 = x0 + _D[s][0], y0 + _D[s][1]
        if img[y1, x1] != 0 or s == s_end:
            break
    if s == s_end:                               # single pixel
        img[y0, x0] = -nbd
        return [(x0, y0)]
    pts = []
    x3, y3 = x0, y0
    prev_s = s ^ 4
    while True:
        s_end = s
        while True:                              # next non-zero neighbour, counter-clockwise from the one after the previous pixel
            s += 1
            x4, y4 = x3 + _D[s & 7][0], y3 + _D[s & 7][1]
            if img[y4, x4] != 0:
                break
        s &= 7
        if 1 <= s <= s_end:                      # the east neighbour was examined and is 0: right-hand exit of the border
            img[y3, x3] = -nbd
        elif img[y3, x3] == 1:
            img[y3, x3] = nbd
        if s != prev_s:
            pts.append((x3, y3))
            prev_s = s
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return pts


def find_contours_external_simple(mask: np.ndarray) -> List[np.ndarray]:
    mask = np.asarray(mask, dtype=bool)
    H, W = mask.shape
    img = np.zeros((H + 2, W + 2), dtype=np.int32)
    img[1:-1, 1:-1] = mask
    out = []
    nbd = 2
    for y in range(1, H + 1):
        lnbd_x, prev = 0, 0
        for x in range(1, W + 2):
            p = int(img[y, x])
            if p != prev:
                if prev == 0 and p == 1 and img[y, lnbd_x] <= 0:      # an outer border starts here and we are not inside a traced one
                    pts = _fetch_outer(img, x, y, nbd)
                    out.append(np.asarray(pts, dtype=np.int32) - 1)   # (remove the frame)
                    p = int(img[y, x])
                prev = p
                if prev not in (0, 1):
                    lnbd_x = x
    return out[::-1]                             # OpenCV hands the contours back last-found first
