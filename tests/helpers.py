"""Shared test helpers: calibrated synthetic weights + comparison utilities.

`make_case` turns the product's seeded synthetic state dict into one with O(1) activations through all ~60
layers on the test frames: every conv gets one scalar gain measured with the oracle on those frames, and the
biased output convs get a target logit spread (so scores and boxes vary and do not saturate). Plain He-init
weights die out (outputs = biases -> an insensitive test) or explode through the residual adds (overflow).
The calibration pass uses the oracle (test infrastructure); the product never sees it - it just receives a
state dict and frames."""
from __future__ import annotations

import functools
import os
import sys
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.yolov10_oracle import BN_EPS, Oracle  # noqa: E402
from yolo_puncture_amd.weights import synthetic_state  # noqa: E402


class _CalibOracle(Oracle):
    """Oracle whose conv() rescales each layer the first time it runs: ONE scalar per layer that brings the
    pre-activation RMS to 1 (biased output convs: a target logit spread). A scalar keeps the network
    well-conditioned; per-channel mean/variance normalisation on a handful of frames amplifies tiny
    position-to-position differences into unit variance at every layer and makes the net chaotic
    (fp32-vs-fp64 box differences of 0.1 px instead of 1e-3 px)."""

    OUT_STD = {"one2one_cv2": 2.0, "one2one_cv3": 1.0, "cv4": 1.0, "cv2": 2.0, "cv3": 1.0}

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.done = set()

    def conv(self, x, name, s=1, act=True, g=1, res=None, keep_fp32=False):
        if name not in self.done:
            self.done.add(name)
            w, b = self.w[name]
            k = w.shape[-1]
            y = F.conv2d(x, w, b, stride=s, padding=k // 2, groups=g)
            parts = name.split(".")
            if (name.startswith("model.23.") or name.startswith("model.22.cv")) and parts[-1] == "2" and parts[2] in self.OUT_STD:
                sc = self.OUT_STD[parts[2]] / y.std().clamp_min(1e-6)   # biased nn.Conv2d: keep the configured bias
                self.w[name] = (w * sc, b)
            else:
                rms = y.pow(2).mean().sqrt().clamp_min(1e-6)
                self.w[name] = (w / rms, b / rms)
        return super().conv(x, name, s=s, act=act, g=g, res=res, keep_fp32=keep_fp32)


@functools.lru_cache(maxsize=16)
def _calibrated_state_cached(variant: str, nc: int, seg: bool, seed: int, shape: Tuple[int, int, int]):
    st = synthetic_state(variant, nc, seg, seed=seed, cls_bias=-3.0)
    im = rand_image((shape[0], shape[1], shape[2], 3), seed=seed)
    co = _CalibOracle(st, variant, nc, seg, "fp32")
    with torch.no_grad():
        co.forward(im)
    # export the calibrated *folded* weights back into an unfused state dict: gamma=1, beta=b', mean=0,
    # var=1-eps  =>  fold gives exactly (w', b')
    out: Dict[str, torch.Tensor] = {}
    for name, (w, b) in co.w.items():
        if f"{name}.conv.weight" in st:
            out[f"{name}.conv.weight"] = w.clone()
            c2 = w.shape[0]
            out[f"{name}.bn.weight"] = torch.ones(c2)
            out[f"{name}.bn.bias"] = b.clone()
            out[f"{name}.bn.running_mean"] = torch.zeros(c2)
            out[f"{name}.bn.running_var"] = torch.full((c2,), 1.0 - BN_EPS)
        elif f"{name}.conv.conv.weight" in st:
            # RepVGGDW was merged at fuse time: export the merged 7x7 as `conv`, and a zero 3x3 as `conv1`
            c2 = w.shape[0]
            for sub, ww, bb in ((".conv", w, b), (".conv1", torch.zeros(c2, 1, 3, 3), torch.zeros(c2))):
                out[f"{name}{sub}.conv.weight"] = ww.clone()
                out[f"{name}{sub}.bn.weight"] = torch.ones(c2)
                out[f"{name}{sub}.bn.bias"] = bb.clone()
                out[f"{name}{sub}.bn.running_mean"] = torch.zeros(c2)
                out[f"{name}{sub}.bn.running_var"] = torch.full((c2,), 1.0 - BN_EPS)
        else:
            out[f"{name}.weight"] = w.clone()
            out[f"{name}.bias"] = b.clone()
    assert set(out) == set(st), sorted(set(out) ^ set(st))[:8]
    return out


def make_case(variant: str = "n", nc: int = 80, seg: bool = False, seed: int = 0,
              shape: Tuple[int, int, int] = (2, 96, 128)) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """-> (state dict calibrated ON the returned frames, uint8 frames [B,H,W,3]). The synthetic network is
    only well-conditioned (no saturated scores, O(1) activations) on the frames it was calibrated on."""
    st = _calibrated_state_cached(variant, nc, seg, seed, tuple(shape))
    return {k: v.clone() for k, v in st.items()}, rand_image((shape[0], shape[1], shape[2], 3), seed=seed)


def rand_image(shape, seed: int = 0) -> torch.Tensor:
    """Seeded uint8 [B,H,W,3] test frames with image-like structure (low-frequency fields at several scales,
    a few flat rectangles, mild pixel noise). Pure uniform noise makes deep features position-independent, so
    BN-style calibration on it does not transfer between frames."""
    B, H, W, C = shape
    g = torch.Generator().manual_seed(seed)
    img = torch.zeros(B, C, H, W)
    for cells in (2, 5, 13, 31):
        f = torch.rand(B, C, cells, cells, generator=g)
        img += F.interpolate(f, size=(H, W), mode="bilinear", align_corners=False) / 4.0
    for b in range(B):
        for _ in range(6):
            y0, x0 = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
            hh, ww = int(torch.randint(4, max(5, H // 2), (1,), generator=g)), int(torch.randint(4, max(5, W // 2), (1,), generator=g))
            img[b, :, y0:y0 + hh, x0:x0 + ww] = torch.rand(C, 1, 1, generator=g)
    img += (torch.rand(B, C, H, W, generator=g) - 0.5) * 0.08
    return (img.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1).contiguous()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / (max|b| + tiny): scale-aware error for activation tensors."""
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# ---- YOLOv8-seg / YOLO11-seg (the families the reference's UI offers) ---------------------------------------------------------
@functools.lru_cache(maxsize=16)
def _calibrated_family_cached(family: str, variant: str, nc: int, seed: int, shape: Tuple[int, int, int]):
    from oracle.yolo_seg_oracle import SegOracle
    from yolo_puncture_amd.weights import synthetic_state_family

    class _CalibSeg(_CalibOracle, SegOracle):
        pass

    st = synthetic_state_family(family, variant, nc, seed=seed, cls_bias=-3.0)
    im = rand_image((shape[0], shape[1], shape[2], 3), seed=seed)
    co = _CalibSeg(st, family, variant, nc, "fp32")
    with torch.no_grad():
        co.forward(im, conf=0.5)
    out: Dict[str, torch.Tensor] = {}
    for name, (w, b) in co.w.items():
        if f"{name}.conv.weight" in st:
            out[f"{name}.conv.weight"] = w.clone()
            c2 = w.shape[0]
            out[f"{name}.bn.weight"] = torch.ones(c2)
            out[f"{name}.bn.bias"] = b.clone()
            out[f"{name}.bn.running_mean"] = torch.zeros(c2)
            out[f"{name}.bn.running_var"] = torch.full((c2,), 1.0 - BN_EPS)
        else:
            out[f"{name}.weight"] = w.clone()
            out[f"{name}.bias"] = b.clone()
    for k, v in st.items():
        if k.endswith("dfl.conv.weight"):
            out[k] = v.clone()
    assert set(out) == set(st), sorted(set(out) ^ set(st))[:8]
    return out


def make_case_family(family: str, variant: str = "n", nc: int = 80, seed: int = 0, shape: Tuple[int, int, int] = (2, 96, 128)):
    """-> (calibrated unfused state dict of a `<family><variant>-seg` checkpoint, uint8 frames it was calibrated on)."""
    st = _calibrated_family_cached(family, variant, nc, seed, tuple(shape))
    return {k: v.clone() for k, v in st.items()}, rand_image((shape[0], shape[1], shape[2], 3), seed=seed)


# ---- fp32 tolerance anchored on the reference's own noise floor -----------------------------------------------------------------
NOISE_FACTOR = 2.0


def assert_within_noise_floor(what: str, eng: torch.Tensor, o32: torch.Tensor, o64: torch.Tensor, target: float, factor: float = NOISE_FACTOR,
                              ceiling: float = None) -> float:
    """The reference's CPU path is an fp32 program: run in fp64 (the oracle's `fp64` mode) the same network gives the exact answer up to
    1e-12, and |oracle_fp32 - oracle_fp64| is the error the reference's OWN arithmetic makes on this input - its noise floor. An fp32
    engine that sums in another order is a second sample of that noise; the contract is
        max |engine_fp32 - oracle_fp64|  <=  min(factor * max |oracle_fp32 - oracle_fp64|, ceiling)        (factor = 2, ceiling = 5 * target)
    on the same elements, which must exist (an empty selection proves nothing). `target` is north_star's absolute figure (1e-3 on box /
    mask floats): printed next to the two measured numbers; the absolute `ceiling` keeps a case whose oracle is itself noisy
    (near-cancelling synthetic weights) from passing an arbitrarily large engine error. Returns the engine's error."""
    assert eng.numel() > 0, f"{what}: empty selection - nothing was compared"
    if ceiling is None:
        ceiling = 5.0 * target
    e = float((eng.double() - o64.double()).abs().max())
    f = float((o32.double() - o64.double()).abs().max())
    tiny = 1e-7 * max(float(o64.double().abs().max()), 1e-30)          # (an exact oracle pair, f = 0: the engine may still round once)
    bound = min(max(factor * f, tiny), ceiling)
    print(f"[noise floor] {what}: engine_fp32 vs oracle_fp64 {e:.3e}; oracle_fp32 vs oracle_fp64 {f:.3e} (x{e / max(f, 1e-30):.2f}); "
          f"asserted bound {bound:.3e} (ceiling {ceiling:g}); north_star target {target:g}; {eng.numel()} values")
    assert e <= bound, (what, e, f, bound)
    return e
