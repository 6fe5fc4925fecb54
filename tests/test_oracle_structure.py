"""Gates that pin the oracle's STRUCTURE to the only numbers the reference publishes for this path (the table copied at
README.md:48-53) and to the exact integers of SURVEY.md Appendix A.8 / B, plus unit tests of its post-process pieces."""
import math

import pytest
import torch

from oracle import postprocess_oracle as po
from oracle.yolov10_oracle import (Oracle, count_conv_flops, count_params, expected_state, v10_postprocess)

# README.md:48-53 (params M, GFLOPs)
PUBLISHED = {"n": (2.3, 6.7), "s": (7.2, 21.6), "m": (15.4, 59.1), "b": (19.1, 92.0), "l": (24.4, 120.3), "x": (29.5, 160.4)}
A8_BOTH_HEADS = {"n": 2_775_520, "s": 8_128_272, "x": 31_808_960}
A8_ONE2ONE = {"n": 2_310_608, "s": 7_277_904, "x": 29_539_392}


@pytest.mark.parametrize("v", list("nsmblx"))
def test_params_and_flops_match_published_table(v):
    p = count_params(expected_state(v, 80, False)) / 1e6
    f = count_conv_flops(v) / 1e9
    assert abs(p - PUBLISHED[v][0]) / PUBLISHED[v][0] < 0.02
    assert abs(f - PUBLISHED[v][1]) / PUBLISHED[v][1] < 0.012


@pytest.mark.parametrize("v", list("nsx"))
def test_exact_parameter_integers(v):
    assert count_params(expected_state(v, 80, False, one2many=True)) == A8_BOTH_HEADS[v]
    assert count_params(expected_state(v, 80, False)) == A8_ONE2ONE[v]


def test_seg_addon_flops():
    add = (count_conv_flops("s", seg=True) - count_conv_flops("s")) / 1e9
    assert abs(add - 11.48) < 0.05          # Proto 10.49 + coefficient branches 1.00


def test_topk_tie_rule_and_duplicates():
    scores = torch.zeros(1, 6, 3)
    scores[0, 4] = torch.tensor([0.9, 0.8, 0.1])      # one anchor appears twice (classes 0 and 1)
    scores[0, 1] = torch.tensor([0.5, 0.5, 0.5])      # exact ties inside an anchor -> class ascending
    scores[0, 2] = torch.tensor([0.5, 0.0, 0.0])      # ties across anchors -> anchor (stage-1 rank) ascending
    boxes = torch.arange(24, dtype=torch.float32).view(1, 6, 4)
    det, idx = v10_postprocess(boxes, scores, max_det=5)
    assert idx[0].tolist() == [4, 4, 1, 1, 1]
    assert det[0, :, 5].tolist() == [0, 1, 0, 1, 2]
    assert torch.equal(det[0, 0, :4], boxes[0, 4])
    # fewer anchors than max_det: k = A
    det, idx = v10_postprocess(boxes, scores, max_det=300)
    assert det.shape == (1, 6, 6)
    assert (det[0, :-1, 4] >= det[0, 1:, 4]).all()


def test_letterbox_geometry_examples():
    g = po.letterbox_geometry(720, 1280)
    assert (g["out_h"], g["out_w"], g["top"], g["bottom"]) == (384, 640, 12, 12)
    g = po.letterbox_geometry(1080, 810)
    assert (g["out_h"], g["out_w"]) == (640, 480)
    g = po.letterbox_geometry(640, 640)
    assert (g["out_h"], g["out_w"], g["top"], g["left"]) == (640, 640, 0, 0)


def test_scale_boxes_inverts_letterbox():
    h0, w0 = 720, 1280
    g = po.letterbox_geometry(h0, w0)
    b0 = torch.tensor([[100., 50., 900., 700.], [0., 0., 1280., 720.]])
    b1 = b0 * g["r"] + torch.tensor([g["left"], g["top"], g["left"], g["top"]])
    back = po.scale_boxes((g["out_h"], g["out_w"]), b1, (h0, w0))
    assert torch.allclose(back, b0, atol=1e-3)
    assert po.conf_filter(torch.tensor([[0, 0, 1, 1, 0.25, 0], [0, 0, 1, 1, 0.2500001, 0]]), 0.25).shape[0] == 1   # strict >


def test_auto_segment_overwrite_and_suppression():
    m = torch.zeros(3, 20, 20)
    m[0, 0:15, 0:15] = 1
    m[1, 10:12, 10:12] = 1           # area 4 -> suppressed when suppress_small
    m[2, 5:18, 5:18] = 1
    ids, info = po.auto_segment_oracle(m, torch.tensor([.9, .8, .7]), torch.tensor([3., 4., 5.]), (20, 20), True, 100)
    assert [i[0] for i in info] == [1, 2] and [i[2] for i in info] == [3, 5]        # ids consecutive over KEPT masks
    assert ids[0, 0] == 1 and ids[6, 6] == 2 and ids[16, 16] == 2 and ids[19, 19] == 0   # later overwrites earlier
    ids2, info2 = po.auto_segment_oracle(m, torch.tensor([.9, .8, .7]), torch.tensor([3., 4., 5.]), (20, 20), False, 100)
    assert len(info2) == 3 and ids2[10, 10] == 3
    ids3, info3 = po.auto_segment_oracle(None, torch.zeros(0), torch.zeros(0), (4, 4), True)
    assert ids3.sum() == 0 and info3 == []


def test_bf16emu_rounds_every_materialised_tensor():
    from helpers import make_case
    st, im = make_case("n", 80, False, 0, (1, 64, 64))
    seen = {}
    Oracle(st, "n", 80, False, "bf16emu", tap=lambda n, x: seen.__setitem__(n, x)).forward(im)
    for n, x in seen.items():
        if n.endswith(".2") and n.startswith("model.23."):
            continue                                    # head logits stay fp32 by design
        assert torch.equal(x, x.to(torch.bfloat16).float()), n


# ---- YOLOv8-seg / YOLO11-seg: published model summaries pin the restated structure ------------------------------------------------
@pytest.mark.parametrize("family,variant,params", [("v8", "n", 3409968), ("11", "n", 2876848)])
def test_seg_families_match_published_parameter_counts_exactly(family, variant, params):
    """ultralytics prints 'YOLOv8n-seg summary: ... 3,409,968 parameters' / 'YOLO11n-seg summary: ... 2,876,848 parameters'"""
    from oracle.yolo_seg_oracle import count_params
    assert count_params(family, variant) == params


@pytest.mark.parametrize("family,variant,millions", [("v8", "s", 11.8), ("v8", "m", 27.3), ("v8", "l", 46.0), ("v8", "x", 71.8),
                                                     ("11", "s", 10.1), ("11", "m", 22.4), ("11", "l", 27.6), ("11", "x", 62.1)])
def test_seg_families_match_the_docs_table(family, variant, millions):
    """the params (M) column of the ultralytics segmentation tables (YOLO11l-seg is listed as 27.6: 27.68 truncated)"""
    from oracle.yolo_seg_oracle import count_params
    assert abs(count_params(family, variant) / 1e6 - millions) < 0.09


def test_nms_oracle_against_brute_force():
    """nms_greedy (torchvision.ops.nms restated) == the definition: walk boxes by descending score, keep a box iff its IoU with every
    box kept so far is <= thr"""
    import torch
    from oracle.yolo_seg_oracle import nms_greedy
    g = torch.Generator().manual_seed(0)
    for n in (1, 7, 200):
        xy = torch.rand(n, 2, generator=g) * 100
        wh = torch.rand(n, 2, generator=g) * 40 + 2
        boxes = torch.cat((xy, xy + wh), 1)
        scores = torch.rand(n, generator=g)
        keep = nms_greedy(boxes, scores, 0.5).tolist()
        order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
        want = []
        for i in order:
            ok = True
            for j in want:
                x1, y1 = max(boxes[i, 0], boxes[j, 0]), max(boxes[i, 1], boxes[j, 1])
                x2, y2 = min(boxes[i, 2], boxes[j, 2]), min(boxes[i, 3], boxes[j, 3])
                inter = max(x2 - x1, 0) * max(y2 - y1, 0)
                a = (boxes[i, 2] - boxes[i, 0]) * (boxes[i, 3] - boxes[i, 1]) + (boxes[j, 2] - boxes[j, 0]) * (boxes[j, 3] - boxes[j, 1]) - inter
                if inter / a > 0.5:
                    ok = False
                    break
            if ok:
                want.append(i)
        assert keep == want


@pytest.mark.parametrize("family,variant", [("v8", "n"), ("v8", "x"), ("11", "n"), ("11", "m"), ("11", "x")])
def test_engine_graph_equals_oracle_layout_for_seg_families(family, variant):
    """the C++ graph builder (yp_create with family) and the oracle's expected_state agree on every parameter name and shape"""
    from oracle.yolo_seg_oracle import expected_state
    from yolo_puncture_amd.engine import Engine
    from yolo_puncture_amd.weights import fold_state, guess_family, guess_variant_family, synthetic_state_family
    st = synthetic_state_family(family, variant, 80, 0)
    assert {k: tuple(v.shape) for k, v in st.items()} == dict(expected_state(family, variant, 80))
    assert guess_family(st) == family and guess_variant_family(st, family) == variant
    e = Engine(variant, 80, True, "bf16", 0, family=family)
    exp = dict(e.expected_weights())
    got = {}
    for k, (w, b) in fold_state(st).items():
        got[k + ".weight"], got[k + ".bias"] = tuple(w.shape), tuple(b.shape)
    assert exp == got
    assert e.plan(1, 640, 640)[-1]["kernel"] == "head_nms_kernel"
    assert e.lib.yp_debug_host_selftest(e._h) > 0
    e.close()
