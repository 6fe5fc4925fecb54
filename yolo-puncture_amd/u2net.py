"""U^2-Net-P on the MI355X engine: host mirror of the reference's `load_unet` / `unet_predict`
(yolo_seg/tasks/unet_segment.py:32-73, called per frame at yolo_seg/app.py:46,184) over libyolop.so's yp_u2net_* entry points.

The network itself (yolo_seg/tasks/models/U2Net.py:424-526: six RSU encoder stages, five decoder stages, six side outputs, 1x1
fusion, sigmoid) runs only through the HIP library: dilated 3x3 convolutions with the folded BatchNorm + ReLU (+ the block
residual) in the epilogue on the matrix cores, ceil-mode 2x2 max-pool, bilinear resize-to-size written straight into the concat
buffer, and one tail kernel for side maps -> fusion -> sigmoid -> min-max normalisation -> mask. There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import YolopError, load_library

BN_EPS = 1e-5

# (kind, in, mid, out) per stage - U2NETP (U2Net.py:429-448), U2NET (:323-342)
_CFG = {
    "p": dict(enc=[("RSU7", 3, 16, 64), ("RSU6", 64, 16, 64), ("RSU5", 64, 16, 64), ("RSU4", 64, 16, 64), ("RSU4F", 64, 16, 64), ("RSU4F", 64, 16, 64)],
              dec=[("RSU4F", 128, 16, 64), ("RSU4", 128, 16, 64), ("RSU5", 128, 16, 64), ("RSU6", 128, 16, 64), ("RSU7", 128, 16, 64)],
              side=[64, 64, 64, 64, 64, 64]),
    "f": dict(enc=[("RSU7", 3, 32, 64), ("RSU6", 64, 32, 128), ("RSU5", 128, 64, 256), ("RSU4", 256, 128, 512), ("RSU4F", 512, 256, 512), ("RSU4F", 512, 256, 512)],
              dec=[("RSU4F", 1024, 256, 512), ("RSU4", 1024, 128, 256), ("RSU5", 512, 64, 128), ("RSU6", 256, 32, 64), ("RSU7", 128, 16, 64)],
              side=[64, 64, 128, 256, 512, 512]),
}


def conv_specs(variant: str = "p") -> List[Tuple[str, int, int, bool]]:
    """(module name, cin, cout, has_bn) of every convolution, in state-dict order of the reference modules."""
    cfg = _CFG[variant]
    out: List[Tuple[str, int, int, bool]] = []

    def rsu(p, kind, cin, mid, co):
        out.append((f"{p}.rebnconvin", cin, co, True))
        out.append((f"{p}.rebnconv1", co, mid, True))
        n = 4 if kind == "RSU4F" else int(kind[3:])
        for i in range(2, n + 1):
            out.append((f"{p}.rebnconv{i}", mid, mid, True))
        for i in range(n - 1, 1, -1):
            out.append((f"{p}.rebnconv{i}d", 2 * mid, mid, True))
        out.append((f"{p}.rebnconv1d", 2 * mid, co, True))

    for i, (kind, cin, mid, co) in enumerate(cfg["enc"]):
        rsu(f"stage{i + 1}", kind, cin, mid, co)
    for j, (kind, cin, mid, co) in enumerate(cfg["dec"]):
        rsu(f"stage{5 - j}d", kind, cin, mid, co)
    for k, c in enumerate(cfg["side"]):
        out.append((f"side{k + 1}", c, 1, False))
    out.append(("outconv", 6, 1, False))
    return out


def synthetic_state(variant: str = "p", seed: int = 0, gain: float = 1.25, side_gain: float = 8.0) -> Dict[str, torch.Tensor]:
    """Seeded state dict in the reference's layout (no checkpoint exists offline): scaled-normal conv weights, small conv biases,
    BatchNorm statistics away from the identity so that the fold is exercised. The gains keep the activations O(1) through the
    ~110 convolutions and spread the fused logit over about +-5 (neither dead nor saturated)."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    for name, cin, cout, bn in conv_specs(variant):
        k = 1 if name == "outconv" else 3
        fan = cin * k * k
        if bn:
            st[f"{name}.conv_s1.weight"] = torch.randn(cout, cin, k, k, generator=g) * (gain / fan) ** 0.5
            st[f"{name}.conv_s1.bias"] = (torch.rand(cout, generator=g) - 0.5) * 0.2
            st[f"{name}.bn_s1.weight"] = 0.6 + 0.8 * torch.rand(cout, generator=g)
            st[f"{name}.bn_s1.bias"] = (torch.rand(cout, generator=g) - 0.5) * 0.4
            st[f"{name}.bn_s1.running_mean"] = torch.randn(cout, generator=g) * 0.1
            st[f"{name}.bn_s1.running_var"] = 0.5 + torch.rand(cout, generator=g)
            st[f"{name}.bn_s1.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        else:
            st[f"{name}.weight"] = torch.randn(cout, cin, k, k, generator=g) * (side_gain / fan) ** 0.5
            st[f"{name}.bias"] = (torch.rand(cout, generator=g) - 0.5) * 0.2
    return st


def fold_state(state: Dict[str, torch.Tensor], variant: str = "p") -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """Conv + eval-mode BatchNorm -> one (weight, bias) per convolution (fp64 fold, fp32 out); plain convs pass through."""
    out = {}
    for name, cin, cout, bn in conv_specs(variant):
        if bn:
            w = state[f"{name}.conv_s1.weight"].double()
            b = state[f"{name}.conv_s1.bias"].double()
            s = state[f"{name}.bn_s1.weight"].double() / torch.sqrt(state[f"{name}.bn_s1.running_var"].double() + BN_EPS)
            out[name] = ((w * s[:, None, None, None]).float(), ((b - state[f"{name}.bn_s1.running_mean"].double()) * s + state[f"{name}.bn_s1.bias"].double()).float())
        else:
            out[name] = (state[f"{name}.weight"].float(), state[f"{name}.bias"].float())
    return out


def _declare(lib: C.CDLL) -> None:
    if getattr(lib, "_u2_declared", False):
        return
    vp = C.c_void_p
    lib.yp_u2net_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.yp_u2net_destroy.argtypes = [vp]
    lib.yp_u2net_weight_count.argtypes = [vp]
    lib.yp_u2net_weight_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    lib.yp_u2net_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), C.c_int]
    lib.yp_u2net_finalize.argtypes = [vp]
    lib.yp_u2net_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.yp_u2net_set_graph.argtypes = [vp, C.c_int]
    lib.yp_u2net_set_graph.restype = C.c_int
    lib.yp_u2net_tensor_count.argtypes = [vp]
    lib.yp_u2net_tensor_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    lib.yp_u2net_tensor_read.argtypes = [vp, C.c_int, vp]
    for fn in ("yp_u2net_create", "yp_u2net_destroy", "yp_u2net_weight_count", "yp_u2net_weight_info", "yp_u2net_set_weight",
               "yp_u2net_finalize", "yp_u2net_forward", "yp_u2net_tensor_count", "yp_u2net_tensor_info", "yp_u2net_tensor_read"):
        getattr(lib, fn).restype = C.c_int
    lib._u2_declared = True


class U2NetEngine:
    """One engine per GPU. dtype 'fp32' = exact fp32 FMA chains on the matrix cores (parity mode, the default: the reference runs
    this network in fp32, unet_segment.py:53-60), 'bf16' = bf16 storage with fp32 accumulation."""

    def __init__(self, variant: str = "p", dtype: str = "fp32", device: int = 0, state: Optional[Dict[str, torch.Tensor]] = None):
        self.lib = load_library()
        _declare(self.lib)
        self.variant, self.device_index = variant, int(device)
        self._h = C.c_void_p()
        self._chk(self.lib.yp_u2net_create(ord(variant), {"bf16": 0, "fp32": 1, "f32": 1}[dtype], self.device_index, C.byref(self._h)))
        if state is not None:
            self.load_state(state)
            self.finalize()

    def _chk(self, rc: int) -> int:
        if rc < 0:
            raise YolopError(f"libyolop error {rc}: {self.lib.yp_last_error().decode()}")
        return rc

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.yp_u2net_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def expected_weights(self) -> List[Tuple[str, Tuple[int, ...]]]:
        n = self._chk(self.lib.yp_u2net_weight_count(self._h))
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        out = []
        for i in range(n):
            self._chk(self.lib.yp_u2net_weight_info(self._h, i, name, 256, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        return out

    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        for name, (w, b) in fold_state(state, self.variant).items():
            for suffix, t in ((".weight", w), (".bias", b)):
                t = t.detach().to(torch.float32).contiguous().cpu()
                shp = (C.c_int64 * t.dim())(*t.shape)
                self._chk(self.lib.yp_u2net_set_weight(self._h, (name + suffix).encode(), C.c_void_p(t.data_ptr()), shp, t.dim()))

    def finalize(self) -> None:
        self._chk(self.lib.yp_u2net_finalize(self._h))

    def set_graph(self, enable: bool) -> None:
        self._chk(self.lib.yp_u2net_set_graph(self._h, 1 if enable else 0))

    def forward(self, im_bgr: torch.Tensor, want_mask: bool = True):
        """im_bgr uint8 cuda [B,H,W,3] (BGR, as cv2 frames are) -> (prob float32 [B,H,W] = sigmoid(d0), norm float32 [B,H,W] = normPRED(prob)
        over the whole call, mask uint8 [B,H,W] in {0,255} = norm > 0.5)."""
        if not (im_bgr.is_cuda and im_bgr.dtype == torch.uint8 and im_bgr.dim() == 4 and im_bgr.shape[-1] == 3):
            raise TypeError("forward expects a uint8 CUDA tensor [B,H,W,3]")
        im_bgr = im_bgr.contiguous()
        B, H, W, _ = im_bgr.shape
        dev = im_bgr.device
        prob = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        norm = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_mask else None
        mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev) if want_mask else None
        self._chk(self.lib.yp_u2net_forward(self._h, C.c_void_p(im_bgr.data_ptr()), B, H, W, C.c_void_p(prob.data_ptr()),
                                            C.c_void_p(norm.data_ptr() if norm is not None else None),
                                            C.c_void_p(mask.data_ptr() if mask is not None else None),
                                            C.c_void_p(int(torch.cuda.current_stream(dev).cuda_stream))))
        self._last = im_bgr
        return prob, norm, mask

    def tensors(self) -> List[dict]:
        n = self._chk(self.lib.yp_u2net_tensor_count(self._h))
        name = C.create_string_buffer(256)
        dims = (C.c_int * 4)()
        return [dict(index=i, name=(self._chk(self.lib.yp_u2net_tensor_info(self._h, i, name, 256, dims)), name.value.decode())[1], shape=tuple(dims))
                for i in range(n)]

    def read_tensor(self, name: str) -> torch.Tensor:
        """Debug tap: NHWC fp32 host copy of an activation of the last forward."""
        for t in self.tensors():
            if t["name"] == name:
                out = torch.empty(t["shape"], dtype=torch.float32)
                self._chk(self.lib.yp_u2net_tensor_read(self._h, t["index"], C.c_void_p(out.data_ptr())))
                return out
        raise KeyError(name)


# ---- the reference's two entry points (yolo_seg/tasks/unet_segment.py:32-73) ------------------------------------------------------------
def load_unet(model_name: str = "u2netp", model_dir: str = "", device="cuda", dtype: str = "fp32") -> U2NetEngine:
    """`load_unet(model_name, model_dir, device)`: model_dir is the path of the `.pth` state dict (as in the reference, :43-46)."""
    variant = {"u2netp": "p", "u2net": "f"}.get(model_name)
    if variant is None:
        raise ValueError(f"unknown model_name {model_name!r} (u2net | u2netp)")
    if not os.path.isfile(model_dir):
        raise FileNotFoundError(f"{model_dir}: no such state dict (nothing is downloaded)")
    state = torch.load(model_dir, map_location="cpu", weights_only=True)
    d = torch.device(device if device != "cuda" else "cuda:0")
    if d.type != "cuda":
        raise ValueError("this engine runs on MI355X GPUs only")
    return U2NetEngine(variant, dtype, d.index or 0, state=state)


def unet_predict(model: U2NetEngine, image: np.ndarray, device="cuda") -> np.ndarray:
    """`unet_predict(model, image)` (:53-73): BGR uint8 HWC frame -> uint8 mask {0,255} [H,W] (numpy, as the reference returns)."""
    if image.ndim != 3 or image.shape[2] != 3 or image.dtype != np.uint8:
        raise TypeError("unet_predict expects a BGR uint8 HWC frame")
    x = torch.from_numpy(np.ascontiguousarray(image)).to(torch.device("cuda", model.device_index))[None]
    _, _, mask = model.forward(x)
    return mask[0].cpu().numpy()
