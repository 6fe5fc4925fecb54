"""CPU ORACLE (test infrastructure, NOT product code): U^2-Net / U^2-Net-P forward and the app's `unet_predict`.

Restates, function by function, the in-tree reference model /root/reference/yolo_seg/tasks/models/U2Net.py (REBNCONV :6-19,
_upsample_like :22-26, RSU7 :30-101, RSU6 :104-166, RSU5 :169-220, RSU4 :223-266, RSU4F :269-314, U2NET :318-420,
U2NETP :424-526) and yolo_seg/tasks/unet_segment.py (normPRED :24-30, unet_predict :53-73) as plain functional torch on a state
dict - no nn.Module of the reference is imported here, so this file travels to the GPU box.

PARITY PINNED: tests/golden/u2netp_*.npz were produced by the REFERENCE module itself (tests/golden/make_u2netp_golden.py imports
U2Net.py in the build container and runs it on seeded weights and frames); tests/test_u2net_oracle.py checks this restatement
against those files. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5          # nn.BatchNorm2d default (U2Net.py:11)

# (kind, in, mid, out) per stage: U2NETP (U2Net.py:429-448) and U2NET (:323-342)
CFG = {
    "p": dict(enc=[("RSU7", 3, 16, 64), ("RSU6", 64, 16, 64), ("RSU5", 64, 16, 64), ("RSU4", 64, 16, 64), ("RSU4F", 64, 16, 64), ("RSU4F", 64, 16, 64)],
              dec=[("RSU4F", 128, 16, 64), ("RSU4", 128, 16, 64), ("RSU5", 128, 16, 64), ("RSU6", 128, 16, 64), ("RSU7", 128, 16, 64)],
              side=[64, 64, 64, 64, 64, 64]),
    "f": dict(enc=[("RSU7", 3, 32, 64), ("RSU6", 64, 32, 128), ("RSU5", 128, 64, 256), ("RSU4", 256, 128, 512), ("RSU4F", 512, 256, 512), ("RSU4F", 512, 256, 512)],
              dec=[("RSU4F", 1024, 256, 512), ("RSU4", 1024, 128, 256), ("RSU5", 512, 64, 128), ("RSU6", 256, 32, 64), ("RSU7", 128, 16, 64)],
              side=[64, 64, 128, 256, 512, 512]),
}


class U2NetOracle:
    def __init__(self, state: Dict[str, Tensor], variant: str = "p", tap: Optional[Callable[[str, Tensor], None]] = None, dtype=torch.float32):
        self.s = {k: v.to(dtype) for k, v in state.items() if v.is_floating_point()}
        self.cfg = CFG[variant]
        self.tap = tap

    # REBNCONV (U2Net.py:6-19): conv3x3(padding = dilation = dirate) -> BatchNorm (eval) -> ReLU
    def rebnconv(self, x: Tensor, name: str, dirate: int = 1) -> Tensor:
        s = self.s
        y = F.conv2d(x, s[f"{name}.conv_s1.weight"], s[f"{name}.conv_s1.bias"], padding=dirate, dilation=dirate)
        y = F.batch_norm(y, s[f"{name}.bn_s1.running_mean"], s[f"{name}.bn_s1.running_var"], s[f"{name}.bn_s1.weight"],
                         s[f"{name}.bn_s1.bias"], training=False, eps=BN_EPS)
        y = F.relu(y)
        if self.tap is not None:
            self.tap(name, y)
        return y

    @staticmethod
    def pool(x: Tensor) -> Tensor:                      # nn.MaxPool2d(2, stride=2, ceil_mode=True)
        return F.max_pool2d(x, 2, stride=2, ceil_mode=True)

    @staticmethod
    def up_like(src: Tensor, tar: Tensor) -> Tensor:    # _upsample_like (U2Net.py:22-26): F.upsample(mode='bilinear') = align_corners False
        return F.interpolate(src, size=tar.shape[2:], mode="bilinear", align_corners=False)

    # RSU-n with n in 7,6,5,4 (U2Net.py:30-266): n-1 levels, the deepest conv dilated by 2
    def rsu(self, x: Tensor, p: str, n: int) -> Tensor:
        hxin = self.rebnconv(x, f"{p}.rebnconvin")
        hx = [self.rebnconv(hxin, f"{p}.rebnconv1")]
        for i in range(2, n):
            hx.append(self.rebnconv(self.pool(hx[-1]), f"{p}.rebnconv{i}"))
        top = self.rebnconv(hx[-1], f"{p}.rebnconv{n}", 2)
        d = self.rebnconv(torch.cat((top, hx[-1]), 1), f"{p}.rebnconv{n - 1}d")
        for i in range(n - 2, 0, -1):
            d = self.rebnconv(torch.cat((self.up_like(d, hx[i - 1]), hx[i - 1]), 1), f"{p}.rebnconv{i}d")
        return d + hxin

    # RSU-4F (U2Net.py:269-314): no pooling, dilations 1,2,4,8 / 4,2,1
    def rsu4f(self, x: Tensor, p: str) -> Tensor:
        hxin = self.rebnconv(x, f"{p}.rebnconvin")
        hx1 = self.rebnconv(hxin, f"{p}.rebnconv1", 1)
        hx2 = self.rebnconv(hx1, f"{p}.rebnconv2", 2)
        hx3 = self.rebnconv(hx2, f"{p}.rebnconv3", 4)
        hx4 = self.rebnconv(hx3, f"{p}.rebnconv4", 8)
        hx3d = self.rebnconv(torch.cat((hx4, hx3), 1), f"{p}.rebnconv3d", 4)
        hx2d = self.rebnconv(torch.cat((hx3d, hx2), 1), f"{p}.rebnconv2d", 2)
        hx1d = self.rebnconv(torch.cat((hx2d, hx1), 1), f"{p}.rebnconv1d", 1)
        return hx1d + hxin

    def stage(self, x: Tensor, name: str, kind: str) -> Tensor:
        y = self.rsu4f(x, name) if kind == "RSU4F" else self.rsu(x, name, int(kind[3:]))
        if self.tap is not None:
            self.tap(name, y)
        return y

    def forward(self, x: Tensor) -> Tuple[Tensor, ...]:
        """x float [B,3,H,W] RGB in [0,1] -> (sigmoid(d0), sigmoid(d1), ..., sigmoid(d6)), U2Net.py:455-526."""
        s, enc, dec = self.s, self.cfg["enc"], self.cfg["dec"]
        hs = []
        hx = x
        for i, (kind, _, _, _) in enumerate(enc):                    # stage1..6 with pool12..56 between (:460-479)
            h = self.stage(hx, f"stage{i + 1}", kind)
            hs.append(h)
            if i + 1 < len(enc):
                hx = self.pool(h)
        ds = [hs[5]]                                                  # hx6
        d = hs[5]
        for j, (kind, _, _, _) in enumerate(dec):                    # stage5d..1d (:480-495)
            skip = hs[4 - j]
            d = self.stage(torch.cat((self.up_like(d, skip), skip), 1), f"stage{5 - j}d", kind)
            ds.append(d)
        feats = [ds[5], ds[4], ds[3], ds[2], ds[1], ds[0]]           # hx1d, hx2d, hx3d, hx4d, hx5d, hx6
        sides = []
        for k, f in enumerate(feats):                                # side1..6 (:498-516): conv3x3 -> upsample to d1's size
            dk = F.conv2d(f, s[f"side{k + 1}.weight"], s[f"side{k + 1}.bias"], padding=1)
            if self.tap is not None:
                self.tap(f"side{k + 1}", dk)
            sides.append(dk if k == 0 else self.up_like(dk, sides[0]))
        d0 = F.conv2d(torch.cat(sides, 1), s["outconv.weight"], s["outconv.bias"])     # :518
        return tuple(torch.sigmoid(t) for t in [d0] + sides)         # :520


def numpy2tensor_oracle(frame_bgr: np.ndarray) -> Tensor:
    """yolo_seg/utils/transform.py:15-20: BGR -> RGB, ToTensor (u8 HWC -> float CHW / 255)."""
    rgb = np.ascontiguousarray(frame_bgr[:, :, ::-1])
    return torch.from_numpy(rgb).permute(2, 0, 1).to(torch.float32) / 255.0


def unet_predict_oracle(state: Dict[str, Tensor], image_bgr: np.ndarray, variant: str = "p") -> Tuple[np.ndarray, np.ndarray]:
    """yolo_seg/tasks/unet_segment.py:53-73: first output (the fused map) -> normPRED (min-max over the tensor, :24-30) -> > 0.5 -> 255.
    returns (normalised float map [H,W], uint8 mask [H,W])."""
    x = numpy2tensor_oracle(image_bgr)[None]
    with torch.no_grad():
        d1 = U2NetOracle(state, variant).forward(x)[0]
    pred = d1[:, 0, :, :]
    ma, mi = torch.max(pred), torch.min(pred)
    dn = (pred - mi) / (ma - mi)
    p = dn.squeeze().numpy()
    return p, np.where(p > 0.5, 255, 0).astype(np.uint8)
