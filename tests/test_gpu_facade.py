"""GPU tests of the drop-in surface: `YOLO(path).predict(...)`, the mask tail and `auto_segment`, against the oracle
pipeline on the same frames. fp32 engine mode, so the comparison is tight (box / score floats within 2 x the reference's own fp32-vs-fp64 noise floor; masks: pixels whose logit is
within 1e-4 of zero may differ - the fraction is bounded)."""
import numpy as np
import pytest
import torch

from helpers import assert_within_noise_floor, make_case, nchw_to_nhwc, rand_image
from oracle import postprocess_oracle as po
from oracle.yolov10_oracle import Oracle
from yolo_puncture_amd.weights import read_ultralytics_pt, save_as_ultralytics_pt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    """A calibrated v10-N-seg checkpoint on disk in the ultralytics layout (fp16 storage, as released weights are),
    calibrated on the letterboxed version of the test frame."""
    frame = rand_image((1, 360, 640, 3), seed=7)[0].numpy()              # a 640x360 BGR frame -> letterbox 384x640
    boxed, _ = po.letterbox(frame)
    from helpers import _CalibOracle
    from yolo_puncture_amd.weights import synthetic_state
    st0 = synthetic_state("n", 80, True, seed=3, cls_bias=-1.0)
    co = _CalibOracle(st0, "n", 80, True, "fp32")
    co.forward(torch.from_numpy(boxed[None]))
    st = {}
    for name, (w, b) in co.w.items():
        if f"{name}.conv.weight" in st0:
            c2 = w.shape[0]
            st.update({f"{name}.conv.weight": w, f"{name}.bn.weight": torch.ones(c2), f"{name}.bn.bias": b,
                       f"{name}.bn.running_mean": torch.zeros(c2), f"{name}.bn.running_var": torch.full((c2,), 1 - 1e-3)})
        else:
            st.update({f"{name}.weight": w, f"{name}.bias": b})
    p = str(tmp_path_factory.mktemp("w") / "v10n-seg-calib.pt")
    save_as_ultralytics_pt(st, p)
    return p, frame


def _oracle_predict(path, frame, conf, retina, mode="fp32"):
    st, meta = read_ultralytics_pt(path)
    boxed, _ = po.letterbox(frame)
    o = Oracle(st, meta["variant"], meta["nc"], meta["seg"], mode).forward(torch.from_numpy(boxed[None]))
    det = o["det"][0]
    keep = det[:, 4] > conf
    det = det[keep]
    boxes_in = det[:, :4].clone()
    H, W = boxed.shape[:2]
    oh, ow = frame.shape[:2]
    det = det.clone()
    det[:, :4] = po.scale_boxes((H, W), det[:, :4], (oh, ow))
    cf = o["coeff"][0][keep]
    if retina:
        m = po.process_mask_native(o["proto"][0], cf, det[:, :4], (oh, ow))
    else:
        m = po.process_mask(o["proto"][0], cf, boxes_in, (H, W))
    return det, m


@pytest.mark.parametrize("retina", [True, False])
def test_predict_matches_oracle_pipeline(ckpt, retina):
    from yolo_puncture_amd import YOLO
    path, frame = ckpt
    conf = 0.3
    model = YOLO(path, dtype="fp32")
    assert model.task == "segment"
    res = model.predict(source=frame, conf=conf, retina_masks=retina, device="cuda")
    assert isinstance(res, list) and len(res) == 1
    r = res[0]
    det, masks = _oracle_predict(path, frame, conf, retina)
    n = det.shape[0]
    assert n >= 3, "test case should produce detections"
    b = r.boxes.cpu().numpy()
    assert len(b.cls) == n
    # rows whose score is not a float near-tie with a neighbour keep their place in the order whatever the fp32 summation order
    sc = det[:, 4].numpy()
    gap = np.abs(np.diff(sc))
    clear = np.ones(n, dtype=bool)
    clear[1:] &= gap > 1e-5
    clear[:-1] &= gap > 1e-5
    assert clear.mean() > 0.5
    assert np.array_equal(b.cls[clear], det[:, 5].numpy()[clear])
    # floats: within 2 x the reference's own fp32 noise floor (|oracle_fp32 - oracle_fp64| on the same rows), helpers.assert_within_noise_floor
    det64, _ = _oracle_predict(path, frame, conf, retina, "fp64")
    assert det64.shape[0] == n
    same = torch.from_numpy(clear) & (det64[:, 5].float() == det[:, 5]) & ((det64[:, :4].float() - det[:, :4]).abs().max(1).values < 0.5) & \
        ((torch.from_numpy(b.xyxy) - det[:, :4]).abs().max(1).values < 0.5)
    assert same.float().mean() > 0.5
    assert_within_noise_floor("facade boxes [px]", torch.from_numpy(b.xyxy)[same], det[:, :4][same], det64[:, :4][same], 1e-3)
    assert_within_noise_floor("facade conf", torch.from_numpy(b.conf)[same], det[:, 4][same], det64[:, 4][same], 1e-3, ceiling=1e-4)
    assert len(r.masks) == n and tuple(r.masks.data.shape[1:]) == tuple(masks.shape[1:])
    diff = (r.masks.data.cpu()[same] != masks[same]).float().mean().item()
    assert diff < 2e-4, diff
    poly = r.masks.xy[int(np.argmax(b.conf))]                        # what app.py:95-101 does
    assert poly.dtype == np.float32 and poly.ndim == 2 and poly.shape[1] == 2
    # a PIL source (app.py:49) and a path-free empty result (conf too high -> masks is None, boxes empty)
    from PIL import Image
    r2 = model.predict(source=Image.fromarray(frame[:, :, ::-1].copy()), conf=conf, retina_masks=retina)[0]
    assert np.array_equal(r2.boxes.cpu().numpy().cls, b.cls)
    r3 = model.predict(frame, conf=0.99999, retina_masks=retina)[0]
    assert len(r3.boxes.cls) == 0 and r3.masks is None


def test_auto_segment_matches_reference_semantics(ckpt):
    from yolo_puncture_amd import YOLO, auto_segment
    path, frame = ckpt
    model = YOLO(path, dtype="fp32")
    model.model.to("cuda")                                             # yolo_with_deva.py:129-130, every frame
    conf = 0.9
    det, masks = _oracle_predict(path, frame, conf, True)
    # lower the bar if this synthetic net has nothing above 0.9: the reference hard-codes conf=0.9 (yolo_with_deva.py:51)
    ids, info = auto_segment({"MIN_AREA_THRESHOLD": 100}, frame, model, min_side=0, suppress_small_mask=True)
    want_ids, want_info = po.auto_segment_oracle(masks if len(masks) else None, det[:, 4], det[:, 5], frame.shape[:2], True, 100)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == frame.shape[:2] and ids.is_cuda
    assert (ids.cpu() != want_ids).float().mean().item() < 2e-4
    assert [(i.id, i.category_id) for i in info] == [(a, c) for a, _, c in want_info]
    assert np.allclose([i.score for i in info], [s for _, s, _ in want_info], atol=1e-4)


def test_auto_segment_min_side_480(ckpt):
    """The reference's NORMAL call (yolo_with_deva.py:118,140: min_side = cfg['size'] > 0): a 720x1280 frame is shrunk to 480x853
    (cv2.resize), predicted, and every float mask is resized back to 720x1280 (torchvision F.resize, antialiased bilinear) before
    the float-area test and the `> 0.5` paint. Whole HIP path (device resize, letterbox, network, mask tail, second resize, paint)
    against the oracle pipeline."""
    from yolo_puncture_amd import YOLO, auto_segment
    path, _ = ckpt
    frame = rand_image((1, 720, 1280, 3), seed=7)[0].numpy()
    model = YOLO(path, dtype="fp32")
    h, w = frame.shape[:2]
    scale = 480 / min(h, w)
    small = po.resize_bilinear_u8_cv2(frame, int(w * scale), int(h * scale))          # :45-48
    assert small.shape[:2] == (480, 853)
    det, masks = _oracle_predict(path, small, 0.9, True)
    assert det.shape[0] >= 1, "the synthetic checkpoint should put at least one detection above the reference's conf=0.9"
    ids, info = auto_segment({"MIN_AREA_THRESHOLD": 100}, frame, model, min_side=480, suppress_small_mask=True)
    want_ids, want_info = po.auto_segment_oracle(masks, det[:, 4], det[:, 5], (h, w), True, 100)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (h, w) and ids.is_cuda
    assert (ids.cpu() != want_ids).float().mean().item() < 2e-4      # (mask pixels whose fp32 logit is within 1e-4 of zero may differ)
    assert len(info) == len(want_info) and [i.id for i in info] == [a for a, _, _ in want_info]
    ws = np.asarray([s for _, s, _ in want_info])
    assert np.allclose([i.score for i in info], ws, atol=1e-4)
    gap = np.abs(np.diff(ws))                                         # rows whose score is a float near-tie with a neighbour may swap
    clear = np.ones(len(ws), dtype=bool)
    clear[1:] &= gap > 1e-5
    clear[:-1] &= gap > 1e-5
    assert clear.mean() > 0.5
    assert [i.category_id for i, c in zip(info, clear) if c] == [c_ for (_, _, c_), c in zip(want_info, clear) if c]


def test_id_mask_resized_bit_level(ckpt):
    """yp_id_mask_resized on the engine's own masks: the antialiased bilinear resize (torch's CPU arithmetic restated), float-area
    test and paint must reproduce `auto_segment_oracle` (= torch F.interpolate(antialias=True) + the reference's loop) BIT FOR BIT -
    up-scaling 2:3 (thousands of exact 0.5 ties), down-scaling, odd ratios, n = 0 / 1 / many, with suppression."""
    from yolo_puncture_amd.engine import Engine
    st, im = make_case("n", 80, True, 0, (2, 96, 160))
    eng = Engine("n", 80, True, "fp32", 0, state=st)
    eng.forward(im.cuda())
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(0)
    for b, (oh, ow), (rh, rw) in ((0, (96, 160), (144, 240)), (1, (200, 333), (90, 160)), (1, (90, 161), (271, 97)), (0, (120, 214), (720, 1280)),
                                  (1, (96, 160), (96, 160))):
        for n in (0, 1, 9):
            coeff = torch.randn(n, 32, generator=g)
            boxes = torch.rand(n, 4, generator=g) * torch.tensor([ow / 2, oh / 2, ow / 2, oh / 2]) + torch.tensor([0, 0, ow / 2, oh / 2])
            if n:
                boxes[0] = torch.tensor([0., 0., float(ow), float(oh)])
            m, _, _ = eng.masks(b, coeff.cuda(), boxes.cuda(), (oh, ow), retina=True)
            for min_area in (50, 2000):
                ids, kept = eng.id_mask_resized(b, coeff.cuda(), boxes.cuda(), (oh, ow), (rh, rw), suppress_small=True, min_area=min_area)
                wi, winfo = po.auto_segment_oracle(m.cpu().float() if n else None, torch.ones(n), torch.zeros(n), (rh, rw), True, min_area)
                assert torch.equal(ids.cpu(), wi), ((oh, ow), (rh, rw), n, int((ids.cpu() != wi).sum()))
                assert [k for k in kept.cpu().tolist() if k > 0] == [a for a, _, _ in winfo]
    eng.close()


def test_mask_kernels_bit_level(ckpt):
    """yp_masks alone on the engine's own prototypes: GEMM -> bilinear -> crop -> >0 and the id paint, n = 0, 1, many,
    boxes touching the borders, with and without small-mask suppression."""
    from yolo_puncture_amd.engine import Engine
    st, im = make_case("n", 80, True, 0, (2, 96, 160))
    eng = Engine("n", 80, True, "fp32", 0, state=st)
    out = eng.forward(im.cuda())
    torch.cuda.synchronize()
    proto = eng.proto()                                               # [B,Hp,Wp,32] fp32 host copy
    g = torch.Generator().manual_seed(0)
    for b, (oh, ow) in ((0, (90, 160)), (1, (96, 160)), (1, (200, 333))):
        for n in (0, 1, 7):
            coeff = torch.randn(n, 32, generator=g)
            boxes = torch.rand(n, 4, generator=g) * torch.tensor([ow / 2, oh / 2, ow / 2, oh / 2]) + torch.tensor([0, 0, ow / 2, oh / 2])
            if n:
                boxes[0] = torch.tensor([0., 0., float(ow), float(oh)])
            m, ids, kept = eng.masks(b, coeff.cuda(), boxes.cuda(), (oh, ow), retina=True, want_ids=True, suppress_small=True, min_area=50)
            want = po.process_mask_native(proto[b].permute(2, 0, 1), coeff, boxes, (oh, ow)) if n else torch.zeros(0, oh, ow)
            assert (m.cpu().float() != want).float().mean().item() < 2e-4 if n else m.shape[0] == 0
            wi, winfo = po.auto_segment_oracle(m.cpu().float() if n else None, torch.ones(n), torch.zeros(n), (oh, ow), True, 50)
            assert torch.equal(ids.cpu(), wi)
            assert [k for k in kept.cpu().tolist() if k > 0] == [a for a, _, _ in winfo]
    eng.close()


@pytest.mark.parametrize("h0,w0", [(720, 1280), (1080, 810), (360, 640), (640, 640), (97, 211), (1333, 777), (33, 1000), (480, 640)])
def test_device_letterbox_bit_exact(h0, w0):
    """yp_letterbox (HIP) against the oracle's restatement of LetterBox = cv2.resize(INTER_LINEAR, 8-bit fixed point) +
    copyMakeBorder(114): integer work, so the bar is bit-exact - down-scaling, up-scaling, identity size, odd sizes,
    both padding orientations."""
    from yolo_puncture_amd import hostops
    from yolo_puncture_amd.engine import letterbox_device
    g = torch.Generator().manual_seed(h0 * 10007 + w0)
    img = torch.randint(0, 256, (h0, w0, 3), generator=g, dtype=torch.uint8)
    # structured content too: gradients make coefficient errors visible, noise makes index errors visible
    img[: h0 // 2] = ((torch.arange(w0)[None, :, None] * 255) // max(w0 - 1, 1)).to(torch.uint8).expand(h0 // 2, w0, 3)
    want, geo = po.letterbox(img.numpy())
    assert geo == hostops.letterbox_geometry(h0, w0)
    got = letterbox_device(img.cuda(), geo).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want), int((got != want).sum())


def test_device_letterbox_rejects_bad_arguments():
    from yolo_puncture_amd.engine import letterbox_device, YolopError
    img = torch.zeros((8, 8, 3), dtype=torch.uint8)
    with pytest.raises(ValueError):
        letterbox_device(img, dict(out_h=8, out_w=8, new_h=8, new_w=8, top=0, left=0))          # host tensor
    with pytest.raises(YolopError):
        letterbox_device(img.cuda(), dict(out_h=8, out_w=8, new_h=9, new_w=8, top=0, left=0))   # does not fit


def test_cabi_allgather_single_rank():
    """yp_comm_* / yp_allgather (RCCL through the C-ABI, no torch.distributed): with one rank the gather is the identity;
    the 2..8-rank form is the same call (the driver's scaling runs use the torch.distributed twin in bench.py)."""
    import ctypes as C
    from yolo_puncture_amd.engine import load_library
    lib = load_library()
    uid = (C.c_char * 128)()
    assert lib.yp_comm_unique_id(uid) == 0, lib.yp_last_error().decode()
    comm = C.c_void_p()
    assert lib.yp_comm_create(uid, 0, 1, 0, C.byref(comm)) == 0, lib.yp_last_error().decode()
    send = torch.rand(32, 300, 6, device="cuda")
    recv = torch.zeros_like(send)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.yp_allgather(comm, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel() * 4, C.c_void_p(st)) == 0, \
        lib.yp_last_error().decode()
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    assert lib.yp_comm_destroy(comm) == 0
