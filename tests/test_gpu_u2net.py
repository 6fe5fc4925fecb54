"""U^2-Net-P on the GPU (yp_u2net_*) against (i) fixtures produced by the REFERENCE module itself (tests/golden/u2netp_*.npz, made by
tests/golden/make_u2netp_golden.py from /root/reference/yolo_seg/tasks/models/U2Net.py) and (ii) the oracle on further shapes.
fp32 engine mode = the reference's precision (unet_segment.py runs the net in fp32): north_star's float bound 1e-3 on the maps, the
uint8 mask bit-exact wherever the normalised value is not within 1e-4 of the 0.5 threshold."""
import os

import numpy as np
import pytest
import torch

from helpers import rand_image
from oracle.u2net_oracle import U2NetOracle, unet_predict_oracle
from yolo_puncture_amd.u2net import U2NetEngine, synthetic_state, unet_predict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def eng():
    e = U2NetEngine("p", "fp32", 0, state=synthetic_state("p", 0))
    yield e
    e.close()


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_against_reference_fixtures(eng, tag):
    z = np.load(os.path.join(GOLD, f"u2netp_{tag}.npz"))
    B, H, W = (int(v) for v in z["shape"])
    im = rand_image((B, H, W, 3), seed=int(z["seed"]))
    prob, norm, mask = eng.forward(im.cuda())
    torch.cuda.synchronize()
    err = np.abs(prob.cpu().numpy() - z["d0"]).max()
    print(tag, (B, H, W), "max |prob - reference| =", err)
    assert err < 1e-3                                                          # measured ~1e-6: fp32 summation order only
    s6 = eng.read_tensor("stage6").permute(0, 3, 1, 2).numpy()
    assert np.abs(s6 - z["stage6"]).max() < 1e-3 * max(1.0, np.abs(z["stage6"]).max())
    s1 = eng.read_tensor("dec1.cat")[..., 64:].permute(0, 3, 1, 2)[:, ::16, ::2, ::2].numpy()   # stage1's output lives in the skip half of dec1's concat buffer
    assert np.abs(s1 - z["stage1"]).max() < 1e-3 * max(1.0, np.abs(z["stage1"]).max())
    d1 = torch.sigmoid(eng.read_tensor("side1"))[..., 0][:, ::2, ::2].numpy()
    assert np.abs(d1 - z["d1"]).max() < 1e-3
    # normPRED + threshold, as unet_predict returns it
    mi, ma = float(z["norm_min"]), float(z["norm_max"])
    want_norm = (z["d0"] - mi) / (ma - mi)
    assert np.abs(norm.cpu().numpy() - want_norm).max() < 1e-3
    bits = np.unpackbits(z["mask_bits"])[: B * H * W].reshape(B, H, W).astype(bool)
    near = np.abs(want_norm - 0.5) < 1e-4
    got = mask.cpu().numpy()
    assert set(np.unique(got)) <= {0, 255}
    assert np.array_equal((got > 0)[~near], bits[~near]) and near.mean() < 5e-3


def test_full_u2net_against_reference_fixture():
    """variant 'f' of include/yolop.h = the full U^2-Net (`load_unet("u2net")`, unet_segment.py:36-37; U2Net.py:318-420): 64 ... 512
    channels, same graph shape as U^2-Net-P. Held to a fixture the reference module produced (tests/golden/u2netf_a.npz)."""
    z = np.load(os.path.join(GOLD, "u2netf_a.npz"))
    B, H, W = (int(v) for v in z["shape"])
    im = rand_image((B, H, W, 3), seed=int(z["seed"]))
    e = U2NetEngine("f", "fp32", 0, state=synthetic_state("f", 0))
    prob, norm, mask = e.forward(im.cuda())
    torch.cuda.synchronize()
    err = np.abs(prob.cpu().numpy() - z["d0"]).max()
    print("u2net (full)", (B, H, W), "max |prob - reference| =", err)
    assert err < 1e-3
    s6 = e.read_tensor("stage6").permute(0, 3, 1, 2).numpy()
    assert np.abs(s6 - z["stage6"]).max() < 1e-3 * max(1.0, np.abs(z["stage6"]).max())
    mi, ma = float(z["norm_min"]), float(z["norm_max"])
    want_norm = (z["d0"] - mi) / (ma - mi)
    assert np.abs(norm.cpu().numpy() - want_norm).max() < 1e-3
    bits = np.unpackbits(z["mask_bits"])[: B * H * W].reshape(B, H, W).astype(bool)
    near = np.abs(want_norm - 0.5) < 1e-4
    assert np.array_equal((mask.cpu().numpy() > 0)[~near], bits[~near]) and near.mean() < 5e-3
    e.close()


@pytest.mark.parametrize("shape,seed", [((1, 33, 47, 3), 5), ((3, 128, 96, 3), 6), ((1, 380, 211, 3), 7)])
def test_against_oracle_other_shapes(eng, shape, seed):
    """odd sizes (ceil-mode pools with clipped windows, non-integer up-sampling ratios at every level), batch > 1"""
    im = rand_image(shape, seed=seed)
    st = synthetic_state("p", 0)
    x = im.flip(-1).permute(0, 3, 1, 2).float() / 255.0
    with torch.no_grad():
        want = U2NetOracle(st, "p").forward(x)[0][:, 0]
    prob, norm, mask = eng.forward(im.cuda())
    assert float((prob.cpu() - want).abs().max()) < 1e-3
    mi, ma = want.min(), want.max()                                            # normPRED runs over the whole call (unet_segment.py:24-30)
    wn = (want - mi) / (ma - mi)
    near = (wn - 0.5).abs() < 1e-4
    assert torch.equal((mask.cpu() > 0)[~near], (wn > 0.5)[~near])


def test_unet_predict_surface():
    """the reference's two entry points: load_unet(model_name, model_dir, device) / unet_predict(model, image)"""
    import tempfile
    from yolo_puncture_amd import load_unet
    st = synthetic_state("p", 1)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "u2netp_synth.pth")
        torch.save(st, path)
        model = load_unet(model_name="u2netp", model_dir=path, device="cuda")
        frame = rand_image((1, 380, 380, 3), seed=9)[0].numpy()                # crop_size=380 (yolo_seg/utils/transform.py:22)
        got = unet_predict(model, frame, device="cuda")
        p, want = unet_predict_oracle(st, frame, "p")
        assert got.dtype == np.uint8 and got.shape == (380, 380)
        near = np.abs(p - 0.5) < 1e-4
        assert np.array_equal(got[~near], want[~near])
        with pytest.raises(FileNotFoundError):
            load_unet("u2netp", os.path.join(d, "missing.pth"))
        model.close()


def test_bf16_mode_is_close():
    """bf16 storage / fp32 accumulation: not the parity mode; the fused map stays within a few 1e-2 of the fp32 reference path"""
    st = synthetic_state("p", 0)
    im = rand_image((1, 160, 160, 3), seed=3)
    x = im.flip(-1).permute(0, 3, 1, 2).float() / 255.0
    with torch.no_grad():
        want = U2NetOracle(st, "p").forward(x)[0][:, 0]
    e = U2NetEngine("p", "bf16", 0, state=st)
    prob, _, _ = e.forward(im.cuda())
    err = float((prob.cpu() - want).abs().mean())
    print("bf16 mean |err|", err)
    assert err < 3e-2
    e.close()


def test_graph_replay_equals_eager(eng):
    """the second and later forwards of a shape replay a hipGraph (engine-owned copies of frame and results around it): same bits as
    the eager pass, also for a new frame in a new tensor and new output tensors"""
    im1, im2 = rand_image((1, 200, 168, 3), seed=11).cuda(), rand_image((1, 200, 168, 3), seed=12).cuda()
    eng.set_graph(False)
    e1 = [t.clone() for t in eng.forward(im1)]
    e2 = [t.clone() for t in eng.forward(im2)]
    eng.set_graph(True)
    for _ in range(2):
        for im, want in ((im1, e1), (im2.clone(), e2)):
            got = eng.forward(im)
            torch.cuda.synchronize()
            for a, b in zip(got, want):
                assert torch.equal(a, b)


@pytest.mark.parametrize("small_max,dtype", [("0", "fp32"), ("1000000000", "fp32"), ("2000", "fp32"), ("1000000000", "bf16")])
def test_conv_kernel_choice_forced(small_max, dtype, monkeypatch):
    """The engine times conv_igemm, the K-split small-map kernel (conv_small.hip) and the halo-tile kernel for large maps
    (conv_halo_f32.hip) per layer; here each is forced for every layer it can run (YOLOP_U2_SMALL_MAX, read at create: 0 = conv_igemm
    everywhere, otherwise conv_small up to that many output pixels and the halo kernel above) and held to the reference fixture: fp32 to the 1e-3 bound of the other
    tests (both kernels are fp32 FMA chains, they differ in summation order only), bf16 to the bf16 mode's closeness bound."""
    monkeypatch.setenv("YOLOP_U2_SMALL_MAX", small_max)
    z = np.load(os.path.join(GOLD, "u2netp_b.npz"))
    B, H, W = (int(v) for v in z["shape"])
    im = rand_image((B, H, W, 3), seed=int(z["seed"]))
    e = U2NetEngine("p", dtype, 0, state=synthetic_state("p", 0))
    try:
        prob, _, _ = e.forward(im.cuda())
        torch.cuda.synchronize()
        err = np.abs(prob.cpu().numpy() - z["d0"])
        print(dtype, "small_max", small_max, "max / mean |prob - reference| =", err.max(), err.mean())
        if dtype == "fp32":
            assert err.max() < 1e-3
            s6 = e.read_tensor("stage6").permute(0, 3, 1, 2).numpy()
            assert np.abs(s6 - z["stage6"]).max() < 1e-3 * max(1.0, np.abs(z["stage6"]).max())
        else:
            assert err.mean() < 3e-2
    finally:
        e.close()
