"""Host logic of the product package (no GPU): weight loading/folding, LetterBox & friends against the oracle's
independent restatement, the Results/Boxes/Masks surface, error behaviour of the facade."""
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import make_case
from oracle import postprocess_oracle as po
from oracle.yolov10_oracle import Oracle, expected_state
from yolo_puncture_amd import hostops
from yolo_puncture_amd.weights import (fold_state, guess_variant, read_ultralytics_pt, save_as_ultralytics_pt,
                                       synthetic_state)


@pytest.mark.parametrize("v,seg", [("n", False), ("s", True), ("m", False), ("x", False)])
def test_synthetic_state_has_the_checkpoint_layout(v, seg):
    st = synthetic_state(v, 80, seg)
    exp = dict(expected_state(v, 80, seg))
    assert set(st) == set(exp)
    assert all(tuple(st[k].shape) == exp[k] for k in st)
    assert guess_variant(st) == v


def test_fold_matches_oracle_fuse():
    st, _ = make_case("s", 80, True, 0, (1, 64, 64))
    folded = fold_state(st)
    orc = Oracle(st, "s", 80, True, "fp32")
    assert set(folded) == set(orc.w)
    for k, (w, b) in folded.items():
        assert torch.equal(w, orc.w[k][0]) and torch.equal(b, orc.w[k][1]), k
    assert not any(k.startswith("model.23.cv2.") for k in folded)


def test_pt_reader_dict_form(tmp_path):
    st = synthetic_state("n", 3, True)
    p = str(tmp_path / "w.pt")
    save_as_ultralytics_pt(st, p)
    got, meta = read_ultralytics_pt(p)
    assert meta["variant"] == "n" and meta["seg"] and meta["nc"] == 3
    assert set(got) == set(st) and torch.allclose(got["model.0.conv.weight"], st["model.0.conv.weight"].half().float())
    with pytest.raises(FileNotFoundError):
        read_ultralytics_pt(str(tmp_path / "missing.pt"))


def test_pt_reader_full_module_pickle_without_ultralytics(tmp_path):
    """Released checkpoints pickle the whole nn.Module by class reference. Fabricate one whose classes live
    in a module path that is NOT importable at load time and read it back through the stub unpickler."""
    import sys
    import types
    modname = "ultralytics.nn.modules.fake_for_test"
    for part in ("ultralytics", "ultralytics.nn", "ultralytics.nn.modules", modname):
        sys.modules.setdefault(part, types.ModuleType(part))
    fake = sys.modules[modname]

    class Blob:                                    # mimics nn.Module's pickled __dict__
        pass

    Blob.__module__, Blob.__qualname__ = modname, "Blob"
    fake.Blob = Blob

    def mod(params=None, buffers=None, children=None):
        m = Blob()
        m._parameters, m._buffers, m._modules = dict(params or {}), dict(buffers or {}), dict(children or {})
        return m

    st = synthetic_state("n", 80, False)
    root = mod()
    for k, v in st.items():                        # build the module tree from the dotted names
        parts = k.split(".")
        node = root
        for p_ in parts[:-1]:
            node = node._modules.setdefault(p_, mod())
        (node._buffers if "running" in parts[-1] else node._parameters)[parts[-1]] = v.half()
    root.names = {0: "needle"}
    root.yaml = {"nc": 80}
    path = str(tmp_path / "full.pt")
    torch.save({"model": root, "ema": None, "date": "x"}, path)
    for part in (modname,):
        del sys.modules[part]                      # the class can no longer be imported: only stubs can load it
    got, meta = read_ultralytics_pt(path)
    assert set(got) == set(st) and meta["variant"] == "n" and meta["names"] == {0: "needle"}


def test_letterbox_matches_oracle_restatement():
    rng = np.random.default_rng(0)
    for (h, w) in ((720, 1280), (1080, 810), (333, 517), (640, 640), (64, 48)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a, ga = hostops.letterbox(img)
        b, gb = po.letterbox(img)
        assert a.shape == b.shape and ga["top"] == gb["top"] and ga["left"] == gb["left"]
        assert np.array_equal(a, b)
        assert a.shape[0] % 32 == 0 and a.shape[1] % 32 == 0


def test_scale_boxes_matches_oracle():
    b = torch.tensor([[12.5, 30., 600., 380.], [-5., -5., 700., 500.]])
    assert torch.allclose(hostops.scale_boxes_t((384, 640), b, (720, 1280)), po.scale_boxes((384, 640), b, (720, 1280)))


def test_contours():
    # cv2's order for an outer border: from the raster-first pixel DOWN first (top-left, bottom-left, bottom-right, top-right)
    m = np.zeros((12, 14), bool)
    m[3:8, 4:11] = True
    assert hostops.largest_external_contour(m).tolist() == [[4, 3], [4, 7], [10, 7], [10, 3]]
    m[0, 0] = True                                  # a second, smaller blob: ignored by "largest", listed LAST by "all" (bottom-up order)
    assert hostops.largest_external_contour(m).shape[0] == 4
    assert hostops.mask_polygon(m, "all").tolist() == [[4, 3], [4, 7], [10, 7], [10, 3], [0, 0]]
    assert [c.tolist() for c in hostops.external_contours(m)] == [[[4, 3], [4, 7], [10, 7], [10, 3]], [[0, 0]]]
    assert hostops.largest_external_contour(np.zeros((5, 5), bool)).shape == (0, 2)
    assert hostops.mask_polygon(np.zeros((5, 5), bool), "all").shape == (0, 2)
    one = np.zeros((5, 5), bool)
    one[2, 3] = True
    assert hostops.largest_external_contour(one).tolist() == [[3, 2]]
    # a ring: only the EXTERNAL border
    ring = np.zeros((9, 9), bool)
    ring[1:8, 1:8] = True
    ring[3:6, 3:6] = False
    assert hostops.largest_external_contour(ring).tolist() == [[1, 1], [1, 7], [7, 7], [7, 1]]
    # RETR_EXTERNAL: a blob inside the hole of another blob is not an external contour - even when its border has more points than the
    # ring's four corners ("largest" must not pick it), and "all" must not list it
    nest = np.zeros((20, 24), bool)
    nest[1:19, 1:23] = True
    nest[4:16, 4:20] = False
    nest[7:13, 8:16] = True
    nest[7, 9] = nest[12, 14] = nest[9, 8] = False   # notches: the inner blob's outer border keeps many more points than 4
    inner = nest.copy()
    inner[:4] = inner[16:] = False
    inner[:, :4] = inner[:, 20:] = False
    assert hostops.largest_external_contour(inner).shape[0] > 4          # on its own the inner blob is a contour with many points
    assert hostops.largest_external_contour(nest).tolist() == [[1, 1], [1, 18], [22, 18], [22, 1]]
    assert hostops.mask_polygon(nest, "all").tolist() == [[1, 1], [1, 18], [22, 18], [22, 1]]
    # ... but a blob in an OPEN bay is external (the bay's background reaches the frame)
    bay = nest.copy()
    bay[8:12, 20:23] = False                         # cut the ring open on the right
    assert len(hostops.external_contours(bay)) == 2
    # a ring closed only by diagonal links still encloses (foreground 8-connected, background 4-connected)
    dia = np.zeros((9, 9), bool)
    for y, x in ((1, 4), (2, 3), (3, 2), (4, 1), (5, 2), (6, 3), (7, 4), (6, 5), (5, 6), (4, 7), (3, 6), (2, 5), (4, 4)):
        dia[y, x] = True
    assert len(hostops.external_contours(dia)) == 1


def test_contours_three_statements_agree():
    """hostops.external_contours (connected components + hole filling + a reversed Moore trace) against tests/suzuki_abe.py (Suzuki &
    Abe's raster scan with +/-NBD labels and the external mode's "last labelled pixel on the row is positive" rule, as cv2 runs it):
    same contours, same order, same points, on random masks of every density (nested blobs, one-pixel-wide rings, diagonal links)."""
    from suzuki_abe import find_contours_external_simple
    rng = np.random.default_rng(7)
    n_nested = 0
    for it in range(600):
        H, W = int(rng.integers(3, 15)), int(rng.integers(3, 17))
        m = rng.random((H, W)) < rng.choice([0.15, 0.35, 0.5, 0.65, 0.8])
        if it % 5 == 0 and H >= 9 and W >= 9:          # structure: a ring with something in its hole
            m[:] = False
            m[1:H - 1, 1:W - 1] = True
            m[2 + it % 2:H - 2, 2:W - 2 - it % 3] = rng.random((H - 4 - it % 2, W - 4 - it % 3)) < 0.3
        a, b = hostops.external_contours(m), find_contours_external_simple(m)
        assert len(a) == len(b), (it, m.astype(int))
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (it, m.astype(int), x.tolist(), y.tolist())
        from scipy import ndimage
        n_nested += ndimage.label(m, structure=np.ones((3, 3)))[1] - len(a)
    assert n_nested > 20           # the sample did contain enclosed blobs


def test_sources_and_results_surface():
    from PIL import Image
    from yolo_puncture_amd.predictor import Boxes, Masks, Results
    rgb = np.zeros((4, 6, 3), np.uint8)
    rgb[..., 0] = 200                                # red in RGB
    imgs, _ = hostops.load_sources(Image.fromarray(rgb))
    assert imgs[0][0, 0].tolist() == [0, 0, 200]     # PIL -> BGR
    arr = np.zeros((4, 6, 3), np.uint8)
    assert hostops.load_sources(arr)[0][0] is arr    # ndarray taken as-is (BGR assumed, never "fixed")
    with pytest.raises(TypeError):
        hostops.load_sources(np.zeros((4, 6), np.uint8))
    d = torch.tensor([[10., 20., 30., 60., 0.9, 2.], [0., 0., 8., 8., 0.5, 1.]])
    bx = Boxes(d, (100, 200))
    n = bx.cpu().numpy()
    assert n.cls.shape == (2,) and np.allclose(n.xywhn[0], [0.1, 0.4, 0.1, 0.4]) and int(np.argmax(n.conf)) == 0
    assert len(Boxes(d[:0], (100, 200)).cls) == 0
    mk = Masks(torch.zeros(2, 100, 200), (100, 200))
    mk.data[0, 10:20, 30:50] = 1
    assert len(mk) == 2 and mk.xy[0].dtype == np.float32 and mk.xy[0].shape == (4, 2) and mk.xy[1].shape == (0, 2)
    r = Results(np.zeros((100, 200, 3), np.uint8), bx, None, {0: "a"})
    assert r.masks is None and len(r) == 2 and r.cpu().numpy().boxes.xyxy.shape == (2, 4)


def test_yolo_constructor_errors(tmp_path):
    from yolo_puncture_amd.predictor import YOLO
    with pytest.raises(FileNotFoundError):
        YOLO(str(tmp_path / "nope.pt"))
    bad = str(tmp_path / "bad.pt")
    torch.save({"model": {"foo.weight": torch.zeros(2)}}, bad)
    with pytest.raises(ValueError):
        YOLO(bad)
    y = YOLO("synthetic:n-seg")
    assert y.task == "segment" and y.variant == "n"
    with pytest.raises(ValueError):
        y.model.to("cpu")                            # the engine is GPU-only; nothing falls back
    with pytest.raises(ValueError):
        y.predict(None)


def test_min_area_rect_len():
    """shaft length from a polygon (reference yolo_seg/utils/mask_tools.py:12-22): exact on rectangles with integer corners"""
    # a 10 x 5 rectangle rotated by atan(3/4), corners listed out of order plus interior / edge points
    pts = [(0, 0), (8, 6), (5, 10), (-3, 4), (4, 3), (1, 5), (2, 4)]
    length, ratio = hostops.get_coord_min_rect_len(np.asarray(pts, dtype=np.float32))
    assert abs(length - 10.0) < 1e-9 and abs(ratio - 2.0) < 1e-9
    # axis-aligned box from a traced contour (what masks.xy holds), float coordinates are truncated like the reference does
    m = np.zeros((20, 30), bool)
    m[3:8, 4:21] = True
    poly = hostops.largest_external_contour(m).astype(np.float32) + 0.7
    length, ratio = hostops.get_coord_min_rect_len(poly)
    assert (length, ratio) == (16.0, 4.0)
    # degenerate inputs
    assert hostops.get_coord_min_rect_len(np.zeros((2, 2), np.float32)) == (0, 0)
    assert hostops.get_coord_min_rect_len(np.asarray([(0, 0), (3, 4), (6, 8)], np.float32)) == (10.0, 10.0)    # collinear: width counted as 1
    # a rotation cannot make the rectangle larger than the axis-aligned bounding box
    rng = np.random.default_rng(0)
    cloud = rng.integers(0, 200, size=(300, 2))
    length, ratio = hostops.get_coord_min_rect_len(cloud)
    bw, bh = np.ptp(cloud[:, 0]), np.ptp(cloud[:, 1])
    assert length * (length / ratio) <= bw * bh + 1e-6


def test_masks2segments_merged_form_of_later_8_3_x():
    """`masks2segments(strategy="all")` of the later 8.3.x releases bridges the contours with merge_multi_segment [U]
    (oracle/postprocess_oracle.py restates it line by line); hostops.merge_contours is the product's form. Same vertices in the same order;
    and - what the reference actually consumes, yolo_seg/app.py:101-103 - the same point set, hence the same minimum-area rectangle as the
    plain concatenation."""
    from oracle import postprocess_oracle as po
    from suzuki_abe import find_contours_external_simple
    rng = np.random.default_rng(5)
    multi = 0
    for it in range(300):
        H, W = int(rng.integers(4, 20)), int(rng.integers(4, 24))
        m = rng.random((H, W)) < rng.choice([0.1, 0.25, 0.4, 0.6])
        cs = find_contours_external_simple(m)
        want = po.masks2segments_contours(cs, "all", merged=True)
        got = hostops.mask_polygon(m, "all_merged")
        assert got.dtype == np.int32 and np.array_equal(got.astype(np.float32), want), (it, m.astype(int))
        plain = hostops.mask_polygon(m, "all")
        assert np.array_equal(plain.astype(np.float32), po.masks2segments_contours(cs, "all", merged=False))
        assert np.array_equal(hostops.mask_polygon(m, "largest").astype(np.float32), po.masks2segments_contours(cs, "largest"))
        if len(cs) > 1:
            multi += 1
            assert {tuple(p) for p in got.tolist()} == {tuple(p) for p in plain.tolist()}
            assert hostops.min_area_rect_size(got) == hostops.min_area_rect_size(plain)
            if len(plain) >= 3:      # (the reference returns (0, 0) below three VERTICES, duplicates counted: utils/mask_tools.py:15-16)
                assert hostops.get_coord_min_rect_len(got.astype(np.float32)) == hostops.get_coord_min_rect_len(plain.astype(np.float32))
        else:
            assert np.array_equal(got, plain)
    assert multi > 100
