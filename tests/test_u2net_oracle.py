"""U^2-Net-P: the oracle (oracle/u2net_oracle.py) against fixtures produced by the REFERENCE module itself
(tests/golden/make_u2netp_golden.py ran /root/reference/yolo_seg/tasks/models/U2Net.py in the build container). This is the one
place where the oracle is pinned by the reference's own arithmetic; the GPU tests then compare the HIP path with both."""
import os

import numpy as np
import pytest
import torch

from helpers import rand_image
from oracle.u2net_oracle import U2NetOracle, unet_predict_oracle
from yolo_puncture_amd.u2net import conv_specs, fold_state, synthetic_state

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(tag, prefix="u2netp"):
    z = np.load(os.path.join(GOLD, f"{prefix}_{tag}.npz"))
    B, H, W = (int(v) for v in z["shape"])
    return z, rand_image((B, H, W, 3), seed=int(z["seed"]))


def test_state_layout_equals_reference_module():
    """names and shapes of the synthetic state (= what conv_specs says the network is) equal the reference nn.Module's state dict"""
    z = np.load(os.path.join(GOLD, "u2netp_params.npz"))
    want = {str(k): tuple(int(x) for x in str(s).split(",")) if str(s) else () for k, s in zip(z["names"], z["shapes"])}
    got = {k: tuple(v.shape) for k, v in synthetic_state("p", 0).items()}
    assert got == want
    assert len(conv_specs("p")) == sum(1 for k in want if k.endswith("conv_s1.weight")) + 7      # REBNCONVs + six side convs + outconv


def test_full_u2net_state_layout_equals_reference_module():
    """the full U^2-Net (`load_unet("u2net")`, U2Net.py:318-420), which include/yolop.h exposes as variant 'f'"""
    z = np.load(os.path.join(GOLD, "u2netf_params.npz"))
    want = {str(k): tuple(int(x) for x in str(s).split(",")) if str(s) else () for k, s in zip(z["names"], z["shapes"])}
    got = {k: tuple(v.shape) for k, v in synthetic_state("f", 0).items()}
    assert got == want


@pytest.mark.parametrize("variant,tag", [("p", "a"), ("p", "b"), ("p", "c"), ("p", "d"), ("f", "a")])
def test_oracle_matches_reference_outputs(variant, tag):
    z, im = load_case(tag, "u2netp" if variant == "p" else "u2netf")
    st = synthetic_state(variant, 0)
    x = im.flip(-1).permute(0, 3, 1, 2).float() / 255.0
    taps = {}
    with torch.no_grad():
        d = U2NetOracle(st, variant, tap=lambda n, t: taps.__setitem__(n, t)).forward(x)
    # same arithmetic (F.conv2d + F.batch_norm + ...), so the bar is float noise, far below north_star's 1e-3
    assert np.abs(d[0][:, 0].numpy() - z["d0"]).max() < 1e-5
    assert np.abs(d[1][:, 0, ::2, ::2].numpy() - z["d1"]).max() < 1e-5
    assert np.abs(taps["stage1"][:, ::16, ::2, ::2].numpy() - z["stage1"]).max() < 1e-4
    assert np.abs(taps["stage6"].numpy() - z["stage6"]).max() < 1e-4
    if im.shape[0] == 1:
        p, mask = unet_predict_oracle(st, im[0].numpy(), variant)
        bits = np.unpackbits(z["mask_bits"])[: mask.size].reshape(mask.shape).astype(bool)
        near = np.abs(p - 0.5) < 1e-5
        assert np.array_equal((mask > 0)[~near], bits[~near]) and near.mean() < 1e-3


def test_fold_is_the_eval_mode_batchnorm():
    """the fold the engine receives (fold_state) reproduces conv -> BatchNorm(eval) of the oracle on random input"""
    st = synthetic_state("p", 3)
    f = fold_state(st, "p")
    o = U2NetOracle(st, "p")
    x = torch.randn(1, 64, 9, 11)
    for name, dil in (("stage2.rebnconvin", 1), ("stage5.rebnconv4", 8)):
        cin = st[f"{name}.conv_s1.weight"].shape[1]
        xi = x[:, :cin]
        want = o.rebnconv(xi, name, dil)
        w, b = f[name]
        got = torch.relu(torch.nn.functional.conv2d(xi, w, b, padding=dil, dilation=dil))
        assert float((got - want).abs().max()) < 1e-5
