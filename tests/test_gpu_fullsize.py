"""BASELINE.json's configurations at FULL size on the GPU.

The oracle's trunk needs minutes of CPU at bs 32 / v10-X, so at these sizes the checks are (i) size-independent
properties of the path - determinism, frame-permutation equivariance (bit-exact: an output pixel's arithmetic does not
depend on where its frame sits in the batch), score order, index ranges - and (ii) the oracle's own HEAD (DFL decode,
sigmoid, two-stage top-k, mask tail: milliseconds on the CPU) run on the engine's fp32 logits for the whole batch, which
pins the post-process kernels on all 32 x 8400 x 80 candidates. Config 1 (v10-N, one 810x1080 frame) is small enough for
the full oracle pipeline end to end."""
import numpy as np
import pytest
import torch

from helpers import _CalibOracle, make_case, nchw_to_nhwc, rand_image
from oracle import postprocess_oracle as po
from oracle.yolov10_oracle import Oracle, v10_postprocess
from yolo_puncture_amd.weights import save_as_ultralytics_pt, synthetic_state

pytestmark = pytest.mark.gpu


def _logits(eng, B, H, W, nc):
    """engine's fp32 head logits -> box [B,64,A], cls [B,nc,A] in the oracle's anchor order (P3,P4,P5 row-major)."""
    box, cls = [], []
    for l in range(3):
        b = eng.read_tensor(eng.find_tensor(f"model.23.one2one_cv2.{l}.2"))      # [B,h,w,64]
        c = eng.read_tensor(eng.find_tensor(f"model.23.one2one_cv3.{l}.2"))
        box.append(b.reshape(B, -1, 64).permute(0, 2, 1))
        cls.append(c.reshape(B, -1, nc).permute(0, 2, 1))
    return torch.cat(box, 2).contiguous(), torch.cat(cls, 2).contiguous()


def _soften(st):
    """the gains were calibrated on small frames; at 640x640 the class logits reach +-60 and the scores saturate to exact
    ties at 1.0. Shrinking the last class conv keeps the scores spread (the test wants an ordering problem, not ties)."""
    st = dict(st)
    for l in range(3):
        st[f"model.23.one2one_cv3.{l}.2.weight"] = st[f"model.23.one2one_cv3.{l}.2.weight"] * 0.15
    return st


def _check_head_against_oracle(eng, orc, out, B, H, W, nc):
    bl, cl = _logits(eng, B, H, W, nc)
    shapes = [(H // s, W // s) for s in (8, 16, 32)]
    boxes, scores = orc.decode(bl, cl, shapes)
    want, widx = v10_postprocess(boxes, scores)
    det, idx = out["det"].cpu(), out["idx"].cpu().long()
    k = want.shape[1]
    assert float((det[:, :k, 4] - want[..., 4]).abs().max()) < 1e-6                  # same sigmoid of the same fp32 logit
    gap_ok = torch.ones_like(want[..., 4], dtype=torch.bool)
    gap_ok[:, :-1] &= (want[:, :-1, 4] - want[:, 1:, 4]) > 1e-6                       # rows that are not float near-ties
    gap_ok[:, 1:] &= (want[:, :-1, 4] - want[:, 1:, 4]) > 1e-6
    assert gap_ok.float().mean() > 0.6                                                # the data must pose an ordering problem
    assert torch.equal(idx[:, :k][gap_ok], widx[gap_ok])                              # integer work: bit-exact
    assert torch.equal(det[:, :k, 5][gap_ok], want[..., 5][gap_ok])
    assert float((det[:, :k, :4] - want[..., :4])[gap_ok].abs().max()) < 1e-3        # px, north_star's float bound
    return want, widx


def _properties(eng, im, out):
    det, idx = out["det"].clone(), out["idx"].clone()
    B = det.shape[0]
    s = det[..., 4]
    assert bool((s[:, :-1] >= s[:, 1:]).all()), "scores must be sorted descending"
    assert bool(((idx >= -1) & (idx < 8400)).all()) and bool(torch.isfinite(det).all())
    assert bool(((det[..., 5] >= 0) & (det[..., 5] < 80) & (det[..., 5] == det[..., 5].round())).all())
    eng.set_graph(True)
    for it in range(40):                      # soak: the persistent kernels' counted waits must not depend on DMA landing order
        again = eng.forward(im)
        assert torch.equal(again["det"], det) and torch.equal(again["idx"], idx), f"replay {it} differs: not deterministic"
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    p = eng.forward(im[perm.to(im.device)].contiguous())
    assert torch.equal(p["det"].cpu(), det.cpu()[perm]) and torch.equal(p["idx"].cpu(), idx.cpu()[perm]), \
        "permuting the frames of a batch must permute the detections bit for bit"


@pytest.mark.parametrize("variant,B", [("s", 32), ("x", 8)])
def test_config_2_and_4_full_size(variant, B):
    """config 2: YOLOv10-S 640x640 bs=32 bf16; config 4: YOLOv10-X 640x640 bs=8 (SURVEY 8d synthetic weights/frames)."""
    from yolo_puncture_amd.engine import Engine
    # weights: the seeded synthetic state, layer gains calibrated on a small frame set (helpers.make_case) so that scores
    # and boxes vary at full size too (plain He-init collapses to the biases: every score a tie)
    st, _ = make_case(variant, 80, False, 0, (2, 320, 320))
    st = _soften(st)
    im = rand_image((B, 640, 640, 3), seed=0).cuda()
    eng = Engine(variant, 80, False, "bf16", 0, state=st)
    out = eng.forward(im)
    torch.cuda.synchronize()
    orc = Oracle(st, variant, 80, False, "fp32")
    _check_head_against_oracle(eng, orc, out, B, 640, 640, 80)
    _properties(eng, im, out)
    eng.close()


def test_config_5_seg_full_size():
    """config 5: YOLOv10-S-seg 640x640 bs=32 + per-frame masks (yolo_with_deva.py:51-86): head as above; the mask tail
    (proto x coeff, bilinear, crop, >0, id paint) against the oracle's on the engine's prototypes, retina and not."""
    from yolo_puncture_amd.engine import Engine
    B = 32
    st = _soften(make_case("s", 80, True, 0, (3, 160, 192))[0])
    im = rand_image((B, 640, 640, 3), seed=5).cuda()
    eng = Engine("s", 80, True, "bf16", 0, state=st)
    out = eng.forward(im)
    torch.cuda.synchronize()
    orc = Oracle(st, "s", 80, True, "fp32")
    want, widx = _check_head_against_oracle(eng, orc, out, B, 640, 640, 80)
    proto = eng.proto()                                                  # [B,160,160,32] fp32 host copy
    assert tuple(proto.shape) == (B, 160, 160, 32)
    for b in (0, 17, 31):
        n = 40
        d = out["det"][b, :n]
        cf = out["coeff"][b, :n]
        boxes = d[:, :4].clamp(0, 640)
        for retina, hw in ((True, (720, 1280)), (False, (640, 640))):
            bx = boxes * torch.tensor([hw[1] / 640, hw[0] / 640, hw[1] / 640, hw[0] / 640], device=boxes.device) if retina else boxes
            m, ids, kept = eng.masks(b, cf, bx, hw, retina=retina, want_ids=True, suppress_small=True, min_area=100)
            pm = proto[b].permute(2, 0, 1)
            ref = po.process_mask_native(pm, cf.cpu(), bx.cpu(), hw) if retina else po.process_mask(pm, cf.cpu(), bx.cpu(), hw)
            assert (m.cpu().float() != ref).float().mean().item() < 2e-4
            wi, winfo = po.auto_segment_oracle(m.cpu().float(), torch.ones(n), torch.zeros(n), hw, True, 100)
            assert torch.equal(ids.cpu(), wi)                            # int64 id paint: bit-exact on the same masks
            assert [k for k in kept.cpu().tolist() if k > 0] == [a for a, _, _ in winfo]
    _properties(eng, im, out)
    eng.close()


def test_config_1_v10n_one_frame_end_to_end(tmp_path):
    """config 1: YOLOv10-N, one 810x1080 BGR frame (the shape of ultralytics' bus.jpg, absent offline) through the drop-in
    surface, fp32 mode, against the full oracle pipeline: LetterBox (-> 640x480), network, top-k, conf filter, scale_boxes."""
    from yolo_puncture_amd import YOLO
    frame = rand_image((1, 1080, 810, 3), seed=11)[0].numpy()
    boxed, geo = po.letterbox(frame)
    assert boxed.shape[:2] == (640, 480)
    st0 = synthetic_state("n", 80, False, seed=3, cls_bias=-1.0)
    co = _CalibOracle(st0, "n", 80, False, "fp32")
    co.forward(torch.from_numpy(boxed[None]))
    st = {}
    for name, (w, b) in co.w.items():
        if f"{name}.conv.weight" in st0:
            c2 = w.shape[0]
            st.update({f"{name}.conv.weight": w, f"{name}.bn.weight": torch.ones(c2), f"{name}.bn.bias": b,
                       f"{name}.bn.running_mean": torch.zeros(c2), f"{name}.bn.running_var": torch.full((c2,), 1 - 1e-3)})
        else:
            st.update({f"{name}.weight": w, f"{name}.bias": b})
    path = str(tmp_path / "v10n-calib.pt")
    save_as_ultralytics_pt(st, path)
    from yolo_puncture_amd.weights import read_ultralytics_pt
    st_rt, meta = read_ultralytics_pt(path)
    conf = 0.3
    o = Oracle(st_rt, "n", 80, False, "fp32").forward(torch.from_numpy(boxed[None]))
    det = o["det"][0]
    det = det[det[:, 4] > conf].clone()
    det[:, :4] = po.scale_boxes((640, 480), det[:, :4], (1080, 810))
    model = YOLO(path, dtype="fp32")
    assert model.task == "detect"
    r = model.predict(source=frame, conf=conf, device="cuda")[0]
    b = r.boxes.cpu().numpy()
    assert len(b.cls) == det.shape[0] and det.shape[0] >= 3
    assert np.array_equal(b.cls, det[:, 5].numpy())
    assert np.abs(b.xyxy - det[:, :4].numpy()).max() < 5e-3 and np.abs(b.conf - det[:, 4].numpy()).max() < 1e-4
    assert r.masks is None
    xywhn = b.xywhn                                                       # cls_bbox_dataset_generate.py:52
    assert xywhn.shape == (len(b.cls), 4) and float(xywhn.min()) >= 0.0 and float(xywhn.max()) <= 1.0
