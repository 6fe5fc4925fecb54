"""yp_mask_contours (HIP: bit image in LDS, parallel Moore traces, RETR_EXTERNAL by crossing parity, hull, rotating calipers) against
what the reference does with a mask per frame (yolo_seg/app.py:101-103: masks.xy[best] -> get_coord_min_rect_len; utils/mask_tools.py:12-22).
The CHECKER is test infrastructure only: tests/suzuki_abe.py (Suzuki & Abe's border following as cv2.findContours(RETR_EXTERNAL,
CHAIN_APPROX_SIMPLE) runs it) for the polygon, scipy's qhull + explicit rotation of all points for the rectangle - neither shares code
with the product. The product's own host statement (hostops.mask_polygon, the fallback Masks.xy takes when the kernel declines) is held
to the same outputs as a second assertion. The polygon is integer work and must be identical point for point, in both masks2segments
strategies; the rectangle is float64."""
import numpy as np
import pytest
import torch

from helpers import rand_image
from suzuki_abe import find_contours_external_simple
from yolo_puncture_amd import hostops
from yolo_puncture_amd.engine import mask_contours_device

pytestmark = pytest.mark.gpu


def _blobs(h, w, seed, thr=0.55, cells=9):
    g = torch.Generator().manual_seed(seed)
    f = torch.rand(1, 1, cells, cells, generator=g)
    m = torch.nn.functional.interpolate(f, size=(h, w), mode="bicubic", align_corners=False)[0, 0]
    return (m > thr).numpy().astype(np.uint8)


def _rot_rect(h, w, cx, cy, a, b, ang):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    c, s = np.cos(ang), np.sin(ang)
    u = (xx - cx) * c + (yy - cy) * s
    v = -(xx - cx) * s + (yy - cy) * c
    return ((np.abs(u) <= a) & (np.abs(v) <= b)).astype(np.uint8)


def _brute_min_rect(points):
    """independent form: hull from scipy's qhull, then every hull edge direction by explicit rotation of ALL points"""
    from scipy.spatial import ConvexHull
    p = np.unique(np.asarray(points, dtype=np.float64).reshape(-1, 2), axis=0)
    if len(p) < 3 or np.linalg.matrix_rank(p - p[0]) < 2:
        d = p.max(0) - p.min(0) if len(p) else np.zeros(2)
        # collinear: length = extent along the line
        if len(p) >= 2:
            q = p[np.argsort(p @ (p[-1] - p[0] + 1e-300))]
            return float(np.hypot(*(q[-1] - q[0]))), 0.0
        return 0.0, 0.0
    hv = p[ConvexHull(p).vertices]
    cand = []
    for i in range(len(hv)):
        e = hv[(i + 1) % len(hv)] - hv[i]
        t = np.arctan2(e[1], e[0])
        R = np.array([[np.cos(t), np.sin(t)], [-np.sin(t), np.cos(t)]])
        q = p @ R.T
        w, h = q[:, 0].max() - q[:, 0].min(), q[:, 1].max() - q[:, 1].min()
        cand.append((w * h, max(w, h), min(w, h)))
    best = min(cand)
    return best[1], best[2]


def _assert_rect(rect_row, points, where=None):
    """The device rectangle of `points` against the brute-force form. Several hull edges can give rectangles of the SAME minimal area with
    different sides (a 3-pixel right triangle: 2 x 1 and sqrt2 x sqrt2); which one cv2.minAreaRect reports is decided by float32 rounding
    inside its calipers loop and cannot be pinned without cv2 - the area is asserted always, the sides when the minimum is unique."""
    from scipy.spatial import ConvexHull
    p = np.unique(np.asarray(points, dtype=np.float64).reshape(-1, 2), axis=0)
    L, S = float(rect_row[0]), float(rect_row[1])
    bl, bw = _brute_min_rect(points)
    assert L * S == pytest.approx(bl * bw, rel=1e-9, abs=1e-7), (where, L, S, bl, bw)
    if len(p) < 3 or np.linalg.matrix_rank(p - p[0]) < 2:
        assert L == pytest.approx(bl, rel=1e-9, abs=1e-9) and S == pytest.approx(bw, abs=1e-7), (where, L, S, bl, bw)
        return
    hv = p[ConvexHull(p).vertices]
    areas = []
    for i in range(len(hv)):
        e = hv[(i + 1) % len(hv)] - hv[i]
        t = np.arctan2(e[1], e[0])
        R = np.array([[np.cos(t), np.sin(t)], [-np.sin(t), np.cos(t)]])
        q = p @ R.T
        w, h = q[:, 0].max() - q[:, 0].min(), q[:, 1].max() - q[:, 1].min()
        areas.append((w * h, max(w, h), min(w, h)))
    amin = min(a for a, _, _ in areas)
    shapes = {(round(l, 6), round(s_, 6)) for a, l, s_ in areas if a <= amin * (1 + 1e-9) + 1e-9}
    if len(shapes) == 1:
        assert L == pytest.approx(bl, rel=1e-9, abs=1e-9) and S == pytest.approx(bw, rel=1e-9, abs=1e-7), (where, L, S, bl, bw)
    else:
        assert any(L == pytest.approx(l, abs=1e-5) and S == pytest.approx(s_, abs=1e-5) for l, s_ in shapes), (where, L, S, shapes)


def _cases():
    c = []
    for seed in range(6):
        c.append((f"blobs{seed}", _blobs(96 + 7 * seed, 130 + 11 * seed, seed)))
    c.append(("blobs_720p", _blobs(720, 1280, 42, thr=0.6, cells=13)))
    c.append(("full_720p", np.ones((720, 1280), np.uint8)))
    for k, ang in enumerate((0.0, 0.3, 0.785398, 1.2, 1.5707963)):
        c.append((f"rect{k}", _rot_rect(200, 260, 130.2, 99.7, 70.0, 9.0, ang)))
    m = np.zeros((64, 80), np.uint8)
    c.append(("empty", m.copy()))
    m1 = m.copy(); m1[10, 20] = 1
    c.append(("one_pixel", m1))
    m2 = m.copy(); m2[10, 20:22] = 1
    c.append(("two_pixels", m2))
    m3 = m.copy(); m3[5, 3:70] = 1
    c.append(("hline", m3))
    m4 = m.copy()
    for i in range(40):
        m4[5 + i, 10 + i] = 1
    c.append(("diag_line", m4))
    m5 = m.copy(); m5[8:50, 8:70] = 1; m5[15:40, 15:60] = 0; m5[20:30, 25:40] = 1; m5[24, 30] = 0     # ring, blob inside the hole, hole inside that
    c.append(("nested", m5))
    m6 = m.copy(); m6[0, :] = 1; m6[:, 0] = 1; m6[63, :] = 1; m6[:, 79] = 1                              # frame touching every border
    c.append(("border_frame", m6))
    m7 = (np.indices((64, 80)).sum(0) % 2).astype(np.uint8)                                            # checkerboard: one 8-connected blob, thousands of candidates
    c.append(("checker", m7))
    m8 = m.copy(); m8[::4, ::4] = 1                                                                    # 320 isolated pixels
    c.append(("dots", m8))
    m9 = m.copy(); m9[20:40, 30] = 1; m9[30, 10:60] = 1                                                # a cross: thin arms are traced out and back
    c.append(("cross", m9))
    return c


def _more_cases():
    c = []
    m = np.zeros((64, 80), np.uint8)
    # a ring whose hole holds a blob with a busier border than the ring's own (notches): the old "largest" would have picked the inner blob
    a = m.copy(); a[4:60, 4:76] = 1; a[12:52, 12:68] = 0; a[20:44, 24:56] = 1
    a[20, 26:54:3] = 0; a[43, 27:54:3] = 0; a[22:42:3, 24] = 0
    c.append(("nested_busy_inner", a))
    # three levels: ring, ring in its hole, blob in that ring's hole - only the outermost is external
    b = m.copy(); b[2:62, 2:78] = 1; b[8:56, 8:72] = 0; b[14:50, 14:66] = 1; b[20:44, 20:60] = 0; b[28:36, 30:50] = 1
    c.append(("nested_three_levels", b))
    # a blob in an OPEN bay is external; several separate blobs around it
    d = m.copy(); d[10:54, 10:70] = 1; d[18:46, 18:70] = 0; d[26:38, 30:50] = 1; d[2:6, 2:30] = 1; d[58:62, 40:78] = 1; d[30, 74] = 1
    c.append(("bay_and_fragments", d))
    # a ring closed only by diagonal links (8-connected foreground / 4-connected background) around a dot, next to a free dot
    e = m.copy()
    for y, x in ((10, 20), (11, 19), (12, 18), (13, 17), (14, 18), (15, 19), (16, 20), (15, 21), (14, 22), (13, 23), (12, 22), (11, 21), (13, 20), (13, 40)):
        e[y, x] = 1
    c.append(("diagonal_ring", e))
    # fragmented needle-like mask: a long thin bar broken into pieces of different sizes
    f = _rot_rect(120, 200, 100.0, 60.0, 80.0, 3.0, 0.4)
    f[:, 60:64] = 0; f[:, 118:121] = 0; f[:, 150:152] = 0
    c.append(("fragmented_bar", f.astype(np.uint8)))
    # equal point counts: "largest" takes the first of the bottom-up list, i.e. the LOWER of two identical squares
    g = m.copy(); g[5:15, 5:15] = 1; g[40:50, 30:40] = 1
    c.append(("tie_two_squares", g))
    # more than 64 outer borders (the table's size) in the segmented path: "largest" goes by rounds - the busy blob C inside ring B inside
    # ring A must be dropped (it lies inside TWO borders: parity per border, not summed), then B (inside A), leaving A; dots all around
    h = m.copy(); h[::3, ::3] = 1; h[8:60, 10:74] = 0
    h[10:58, 12:72] = 1; h[14:54, 16:68] = 0; h[18:50, 20:64] = 1; h[22:46, 24:60] = 0; h[28:40, 30:54] = 1
    h[28, 32:52:3] = 0; h[39, 31:52:3] = 0
    c.append(("many_dots_nested", h))
    return c


ALL = _cases() + _more_cases()


def _oracle_polygon(mask, strategy):
    """masks2segments [U] on the test-side contour statement: every external contour bottom-up ("all": concatenated; "largest": the one
    with the most points, the first of the list on a tie) -> (int32 [m,2], contour lengths)."""
    cs = find_contours_external_simple(mask.astype(bool))
    if not cs:
        return np.zeros((0, 2), np.int32), []
    if strategy == "all":
        return np.concatenate(cs).astype(np.int32), [len(c) for c in cs]
    best = max(range(len(cs)), key=lambda i: (len(cs[i]), -i))
    return cs[best].astype(np.int32), [len(cs[best])]


@pytest.mark.parametrize("strategy", ["all", "largest"])
@pytest.mark.parametrize("name,mask", ALL, ids=[n for n, _ in ALL])
def test_contour_and_rect_match_host(name, mask, strategy):
    polys, rect, parts = mask_contours_device(torch.from_numpy(mask)[None].cuda(), max_pts=8192, strategy=strategy, want_parts=True)
    want, want_parts = _oracle_polygon(mask, strategy)                    # test-side statement (tests/suzuki_abe.py)
    got = polys[0]
    if name in ("dots", "many_dots_nested") and strategy == "all":      # hundreds of isolated pixels: more outer borders than the device lists (64) -> declined, not wrong
        assert got is None
        return
    assert got is not None, "the device path must handle this mask"
    assert got.dtype == np.int32 and got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want)                                       # integer work: identical, point for point, same order
    assert parts[0] == want_parts                                          # the list's contours, bottom-up
    if want.shape[0] >= 3:                                                 # rectangle: qhull + rotation of all points (no product code)
        _assert_rect(rect[0], want, name)
    elif want.shape[0] == 2:
        assert rect[0, 0] == pytest.approx(float(np.hypot(*(want[1] - want[0]).astype(np.float64))), abs=1e-12) and rect[0, 1] == pytest.approx(0.0, abs=1e-12)
    else:
        assert rect[0, 0] == 0.0 and rect[0, 1] == 0.0
    # second assertion: the product's host statement (the fallback path of Masks.xy) says the same
    host = hostops.mask_polygon(mask.astype(bool), strategy)
    assert np.array_equal(got, host)
    wl, ww = hostops.min_area_rect_size(host) if host.shape[0] else (0.0, 0.0)
    assert rect[0, 0] == pytest.approx(wl, rel=1e-12, abs=1e-12) and rect[0, 1] == pytest.approx(ww, rel=1e-12, abs=1e-9)


def test_random_masks_against_suzuki_abe():
    """A few hundred random masks of every density (nested blobs, one-pixel rings, diagonal links) in ONE batched device call per strategy,
    each held to tests/suzuki_abe.py + the brute-force rectangle."""
    rng = np.random.default_rng(11)
    H, W, N = 24, 40, 320
    ms = np.zeros((N, H, W), np.uint8)
    for i in range(N):
        h, w = int(rng.integers(3, H + 1)), int(rng.integers(3, W + 1))
        m = rng.random((h, w)) < rng.choice([0.15, 0.35, 0.5, 0.65, 0.8])
        if i % 5 == 0 and h >= 9 and w >= 9:
            m[:] = False
            m[1:h - 1, 1:w - 1] = True
            m[2 + i % 2:h - 2, 2:w - 2 - i % 3] = rng.random((h - 4 - i % 2, w - 4 - i % 3)) < 0.3
        ms[i, :h, :w] = m
    for strategy in ("all", "largest"):
        polys, rect = mask_contours_device(torch.from_numpy(ms).cuda(), strategy=strategy)
        declined = 0
        for i in range(N):
            want, _ = _oracle_polygon(ms[i], strategy)
            if polys[i] is None:                       # (more than 64 outer borders in "all": declined, the host path takes over)
                declined += 1
                assert strategy == "all" and len(find_contours_external_simple(ms[i].astype(bool))) > 64
                continue
            assert np.array_equal(polys[i], want), (i, strategy)
            if want.shape[0] >= 3:
                _assert_rect(rect[i], want, (i, strategy))
        assert declined < N // 10


def test_nested_blob_is_not_external():
    """RETR_EXTERNAL: the busy blob inside the ring's hole has more run end points than the ring's four corners, and on its own it IS the
    contour - inside the hole it must be skipped by both strategies (yolo_seg/app.py:101-103 would otherwise measure the wrong shaft)."""
    name, a = _more_cases()[0]
    inner = a.copy(); inner[:12] = 0; inner[52:] = 0; inner[:, :12] = 0; inner[:, 68:] = 0
    p_in, _ = mask_contours_device(torch.from_numpy(inner)[None].cuda(), strategy="largest")
    assert len(p_in[0]) > 20
    for strategy in ("largest", "all"):
        p, rect = mask_contours_device(torch.from_numpy(a)[None].cuda(), strategy=strategy)
        assert p[0].tolist() == [[4, 4], [4, 59], [75, 59], [75, 4]]
        assert rect[0, 0] == pytest.approx(71.0) and rect[0, 1] == pytest.approx(55.0)
    # the tie rule of "largest": first of the bottom-up list
    g = _more_cases()[5][1]
    p, _ = mask_contours_device(torch.from_numpy(g)[None].cuda(), strategy="largest")
    assert p[0].tolist() == [[30, 40], [30, 49], [39, 49], [39, 40]]


def test_noise_mask_one_lane_path():
    """More than 1024 border starts: one lane per candidate walks its whole border; "largest" then tests its winner against the borders that
    start before it (rounds), "all" declines beyond 64 outer borders."""
    rng = np.random.default_rng(3)
    m = (rng.random((200, 300)) < 0.5).astype(np.uint8)
    m[40:160, 60:240] = 1; m[60:140, 90:210] = 0; m[80:120, 120:180] = (rng.random((40, 60)) < 0.62)     # a noisy blob inside a ring inside noise
    polys, rect = mask_contours_device(torch.from_numpy(m)[None].cuda(), strategy="largest")
    want = _oracle_polygon(m, "largest")[0]
    assert polys[0] is not None and np.array_equal(polys[0], want)
    assert np.array_equal(want, hostops.mask_polygon(m.astype(bool), "largest"))
    polys, _ = mask_contours_device(torch.from_numpy(m)[None].cuda(), strategy="all")
    assert polys[0] is None


def test_batch_of_masks_and_fallback_codes():
    ms = np.stack([_blobs(120, 160, s) for s in range(5)] + [np.zeros((120, 160), np.uint8)])
    polys, rect = mask_contours_device(torch.from_numpy(ms).cuda())
    for i in range(6):
        assert np.array_equal(polys[i], _oracle_polygon(ms[i], "all")[0])
    polys_l, _ = mask_contours_device(torch.from_numpy(ms).cuda(), strategy="largest")
    for i in range(6):
        assert np.array_equal(polys_l[i], _oracle_polygon(ms[i], "largest")[0])
    assert polys[5].shape == (0, 2) and rect[5, 0] == 0.0
    # too many points for the caller's buffer -> the device pass declines (host path takes over in Masks.xy)
    polys, _ = mask_contours_device(torch.from_numpy((np.indices((64, 80)).sum(0) % 2).astype(np.uint8))[None].cuda(), max_pts=16)
    assert polys[0] is None
    # a blob wider than the LDS image (bounding box 2200 x 900 > 126 KB of bits) -> declined, not wrong
    big = np.zeros((900, 2200), np.uint8); big[10:890, 5:2195] = 1
    polys, _ = mask_contours_device(torch.from_numpy(big)[None].cuda())
    assert polys[0] is None


def test_masks_xy_uses_the_device_path_and_matches_host():
    from yolo_puncture_amd.predictor import Masks
    ms = np.stack([_blobs(180, 240, s) for s in range(3)])
    dev = Masks(torch.from_numpy(ms).cuda().float(), (180, 240), u8=torch.from_numpy(ms).cuda())
    host = Masks(torch.from_numpy(ms).float(), (180, 240))
    for a, b in zip(dev.xy, host.xy):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    for i in range(3):
        l, r = dev.min_rect_len(i)
        hl, hr = hostops.get_coord_min_rect_len(host.xy[i])
        assert l == pytest.approx(hl, rel=1e-12) and r == pytest.approx(hr, rel=1e-9)
    # masks at the letterboxed size (retina_masks=False): polygons are scaled to the original frame on the host, as ultralytics does
    dev2 = Masks(torch.from_numpy(ms).cuda().float(), (360, 480), u8=torch.from_numpy(ms).cuda())
    host2 = Masks(torch.from_numpy(ms).float(), (360, 480))
    for a, b in zip(dev2.xy, host2.xy):
        assert np.array_equal(a, b)


def test_hole_border_with_local_tops_is_not_a_contour():
    """The segmented trace treats every local top as a checkpoint, also those on a hole's border (spikes rising from the floor of a hole);
    a border counts only from its raster-first PIXEL, which for a hole border is no local top - so the wiggly hole below must not beat the
    plain outer border (4 corner points), exactly as the host restatement (outer borders only) has it. A second mask nests a blob with its
    own spikes inside the hole."""
    m = np.zeros((80, 120), np.uint8)
    m[10:70, 10:110] = 1
    m[20:60, 20:100] = 0                       # the hole
    for x in range(24, 96, 4):                 # spikes from the hole's floor: tips are local tops (W, NW, N, NE clear)
        m[40 + (x % 8):60, x] = 1
    m2 = m.copy()
    m2[28:36, 40:80] = 1                       # a blob inside the hole
    m2[24:28, 44:76:4] = 1                     # ... with spikes of its own (outer-border local tops of the nested blob)
    polys, rect = mask_contours_device(torch.from_numpy(np.stack([m, m2])).cuda(), strategy="largest")
    for i, mm in enumerate((m, m2)):
        want = _oracle_polygon(mm, "largest")[0]
        assert polys[i] is not None and np.array_equal(polys[i], want), (i, len(polys[i]), len(want))
    assert len(polys[0]) == 4


def test_masks_xy_all_merged_strategy():
    """MASK_POLYGON_STRATEGY = "all_merged" (masks2segments of the later 8.3.x releases [U]): contours from the device, bridged on the host,
    against the oracle's restatement of merge_multi_segment on the test-side contours; the rectangle is the one "all" gives."""
    from oracle import postprocess_oracle as po
    from yolo_puncture_amd.predictor import Masks
    ms = np.stack([_blobs(120, 160, s, thr=0.62) for s in range(4)] + [ALL[[n for n, _ in ALL].index("fragmented_bar")][1][:120, :160]])
    dev = Masks(torch.from_numpy(ms).cuda().float(), (120, 160), u8=torch.from_numpy(ms).cuda(), strategy="all_merged")
    plain = Masks(torch.from_numpy(ms).cuda().float(), (120, 160), u8=torch.from_numpy(ms).cuda(), strategy="all")
    multi = 0
    for i in range(len(ms)):
        cs = find_contours_external_simple(ms[i].astype(bool))
        want = po.masks2segments_contours(cs, "all", merged=True)
        assert np.array_equal(dev.xy[i], want), i
        multi += len(cs) > 1
        if len(plain.xy[i]) >= 3:
            assert dev.min_rect_len(i) == plain.min_rect_len(i)
    assert multi >= 2
