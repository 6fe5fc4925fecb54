"""CPU ORACLE (test infrastructure, NOT product code): YOLOv8-seg and YOLO11-seg - the checkpoints the reference's UI actually
offers (`yolov8n-seg`, `yolo11n-seg`, `yolo11x-seg` fine-tunes, /root/reference/yolo_seg/app.py:218-223; `seg/yolo11n-seg-finetune.pt`,
yolo_seg/yolo_with_deva.py:226) - as a functional torch restatement on a state dict.

PARITY UNPINNED, like yolov10_oracle.py: the arithmetic lives in the un-vendored, un-pinned PyPI package `ultralytics`
(/root/reference/pyproject.toml:23), absent from /root/reference and not installed; the reference holds no fixtures for it. What is
restated here from the public ultralytics sources (cfg/models/v8/yolov8-seg.yaml, cfg/models/11/yolo11-seg.yaml, nn/modules/block.py
C2f / C3k2 / C3k / Bottleneck / SPPF / C2PSA / PSABlock / Attention / Proto, nn/modules/head.py Detect / Segment, nn/tasks.py
parse_model, utils/ops.py non_max_suppression, torchvision.ops.nms) is pinned only by the published model summaries
(parameter counts of yolov8n-seg / yolo11n-seg / yolo11x-seg, tests/test_oracle_structure.py) and by build-made fixtures.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .yolov10_oracle import BN_EPS, MAX_DET, NM, REG_MAX, Oracle, _conv_entries, make_divisible

Tensor = torch.Tensor

# [depth, width, max_channels] (yolov8-seg.yaml / yolo11-seg.yaml `scales`)
SCALES = {
    "v8": {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)},
    "11": {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512), "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512)},
}
MAX_WH = 7680.0      # class offset of the batched NMS (ops.non_max_suppression)
IOU_THRES = 0.7      # predictor default `iou`


def layer_table(family: str, variant: str) -> List[dict]:
    """(from, kind, yaml c2, yaml repeats, extras) per layer index. kinds: conv, c2f, c3k2, sppf, c2psa, up, cat."""
    L: List[dict] = []

    def add(frm, kind, c2=0, n=1, **kw):
        L.append(dict(f=frm, kind=kind, c2=c2, n=n, **kw))

    if family == "v8":                                   # yolov8-seg.yaml
        add(-1, "conv", 64, k=3, s=2)                    # 0
        add(-1, "conv", 128, k=3, s=2)                   # 1
        add(-1, "c2f", 128, 3, shortcut=True)            # 2
        add(-1, "conv", 256, k=3, s=2)                   # 3
        add(-1, "c2f", 256, 6, shortcut=True)            # 4
        add(-1, "conv", 512, k=3, s=2)                   # 5
        add(-1, "c2f", 512, 6, shortcut=True)            # 6
        add(-1, "conv", 1024, k=3, s=2)                  # 7
        add(-1, "c2f", 1024, 3, shortcut=True)           # 8
        add(-1, "sppf", 1024, k=5)                       # 9
        add(-1, "up")                                    # 10
        add([-1, 6], "cat")                              # 11
        add(-1, "c2f", 512, 3, shortcut=False)           # 12
        add(-1, "up")                                    # 13
        add([-1, 4], "cat")                              # 14
        add(-1, "c2f", 256, 3, shortcut=False)           # 15  P3
        add(-1, "conv", 256, k=3, s=2)                   # 16
        add([-1, 12], "cat")                             # 17
        add(-1, "c2f", 512, 3, shortcut=False)           # 18  P4
        add(-1, "conv", 512, k=3, s=2)                   # 19
        add([-1, 9], "cat")                              # 20
        add(-1, "c2f", 1024, 3, shortcut=False)          # 21  P5
    elif family == "11":                                 # yolo11-seg.yaml ; parse_model forces c3k=True for scales m, l, x
        big = variant in "mlx"
        add(-1, "conv", 64, k=3, s=2)                    # 0
        add(-1, "conv", 128, k=3, s=2)                   # 1
        add(-1, "c3k2", 256, 2, c3k=big, e=0.25)         # 2
        add(-1, "conv", 256, k=3, s=2)                   # 3
        add(-1, "c3k2", 512, 2, c3k=big, e=0.25)         # 4
        add(-1, "conv", 512, k=3, s=2)                   # 5
        add(-1, "c3k2", 512, 2, c3k=True, e=0.5)         # 6
        add(-1, "conv", 1024, k=3, s=2)                  # 7
        add(-1, "c3k2", 1024, 2, c3k=True, e=0.5)        # 8
        add(-1, "sppf", 1024, k=5)                       # 9
        add(-1, "c2psa", 1024, 2)                        # 10
        add(-1, "up")                                    # 11
        add([-1, 6], "cat")                              # 12
        add(-1, "c3k2", 512, 2, c3k=big, e=0.5)          # 13
        add(-1, "up")                                    # 14
        add([-1, 4], "cat")                              # 15
        add(-1, "c3k2", 256, 2, c3k=big, e=0.5)          # 16  P3
        add(-1, "conv", 256, k=3, s=2)                   # 17
        add([-1, 13], "cat")                             # 18
        add(-1, "c3k2", 512, 2, c3k=big, e=0.5)          # 19  P4
        add(-1, "conv", 512, k=3, s=2)                   # 20
        add([-1, 10], "cat")                             # 21
        add(-1, "c3k2", 1024, 2, c3k=True, e=0.5)        # 22  P5
    else:
        raise ValueError(family)
    return L


def resolve(family: str, variant: str) -> Tuple[List[int], List[int]]:
    depth, width, maxc = SCALES[family][variant]
    tbl = layer_table(family, variant)
    ch, reps = [], []
    for i, e in enumerate(tbl):
        if e["kind"] == "up":
            c = ch[-1]
        elif e["kind"] == "cat":
            c = sum(ch[j] if j >= 0 else ch[i + j] for j in e["f"])
        else:
            c = make_divisible(min(e["c2"], maxc) * width, 8)
        ch.append(c)
        reps.append(max(round(e["n"] * depth), 1) if e["n"] > 1 else e["n"])
    return ch, reps


def head_info(family: str, variant: str, nc: int):
    """-> (head layer index, (ch P3,P4,P5), c2 box hidden, c3 cls hidden, c4 coeff hidden, npr)."""
    ch, _ = resolve(family, variant)
    idx = (15, 18, 21) if family == "v8" else (16, 19, 22)
    chs = tuple(ch[i] for i in idx)
    _, width, maxc = SCALES[family][variant]
    c2 = max(16, chs[0] // 4, REG_MAX * 4)
    c3 = max(chs[0], min(nc, 100))
    c4 = max(chs[0] // 4, NM)
    npr = make_divisible(min(256, maxc) * width, 8)
    return (22 if family == "v8" else 23), chs, c2, c3, c4, npr


def expected_state(family: str, variant: str, nc: int = 80) -> List[Tuple[str, Tuple[int, ...]]]:
    """names/shapes of the unfused parameters + BN buffers of `<family><variant>-seg` (as `model.state_dict()` lists them, minus
    num_batches_tracked), including the constant DFL conv."""
    tbl = layer_table(family, variant)
    ch, reps = resolve(family, variant)
    E: List[Tuple[str, Tuple[int, ...]]] = []
    for i, e in enumerate(tbl):
        p, kind = f"model.{i}", e["kind"]
        if kind in ("up", "cat"):
            continue
        c1 = 3 if i == 0 else ch[i - 1]
        c2 = ch[i]
        if kind == "conv":
            E += _conv_entries(p, c1, c2, e["k"])
        elif kind == "c2f":
            c, n = int(c2 * 0.5), reps[i]
            E += _conv_entries(f"{p}.cv1", c1, 2 * c, 1)
            E += _conv_entries(f"{p}.cv2", (2 + n) * c, c2, 1)
            for j in range(n):
                E += _conv_entries(f"{p}.m.{j}.cv1", c, c, 3)
                E += _conv_entries(f"{p}.m.{j}.cv2", c, c, 3)
        elif kind == "c3k2":
            c, n = int(c2 * e["e"]), reps[i]
            E += _conv_entries(f"{p}.cv1", c1, 2 * c, 1)
            E += _conv_entries(f"{p}.cv2", (2 + n) * c, c2, 1)
            for j in range(n):
                q = f"{p}.m.{j}"
                if e["c3k"]:                              # C3k(c, c, 2): c_ = c/2, two Bottleneck(c_, c_, k=(3,3), e=1.0)
                    c_ = int(c * 0.5)
                    E += _conv_entries(f"{q}.cv1", c, c_, 1)
                    E += _conv_entries(f"{q}.cv2", c, c_, 1)
                    E += _conv_entries(f"{q}.cv3", 2 * c_, c, 1)
                    for t in range(2):
                        E += _conv_entries(f"{q}.m.{t}.cv1", c_, c_, 3)
                        E += _conv_entries(f"{q}.m.{t}.cv2", c_, c_, 3)
                else:                                     # Bottleneck(c, c, shortcut, g): default e = 0.5
                    c_ = int(c * 0.5)
                    E += _conv_entries(f"{q}.cv1", c, c_, 3)
                    E += _conv_entries(f"{q}.cv2", c_, c, 3)
        elif kind == "sppf":
            c_ = c1 // 2
            E += _conv_entries(f"{p}.cv1", c1, c_, 1)
            E += _conv_entries(f"{p}.cv2", 4 * c_, c2, 1)
        elif kind == "c2psa":
            c = int(c1 * 0.5)
            nh = c // 64
            hd = c // nh
            kd = int(hd * 0.5)
            E += _conv_entries(f"{p}.cv1", c1, 2 * c, 1)
            E += _conv_entries(f"{p}.cv2", 2 * c, c1, 1)
            for j in range(reps[i]):
                q = f"{p}.m.{j}"
                E += _conv_entries(f"{q}.attn.qkv", c, c + 2 * kd * nh, 1)
                E += _conv_entries(f"{q}.attn.proj", c, c, 1)
                E += _conv_entries(f"{q}.attn.pe", c, c, 3, g=c)
                E += _conv_entries(f"{q}.ffn.0", c, 2 * c, 1)
                E += _conv_entries(f"{q}.ffn.1", 2 * c, c, 1)
    hi, chs, c2, c3, c4, npr = head_info(family, variant, nc)
    p = f"model.{hi}"
    for l, x in enumerate(chs):
        E += _conv_entries(f"{p}.cv2.{l}.0", x, c2, 3)
        E += _conv_entries(f"{p}.cv2.{l}.1", c2, c2, 3)
        E += [(f"{p}.cv2.{l}.2.weight", (4 * REG_MAX, c2, 1, 1)), (f"{p}.cv2.{l}.2.bias", (4 * REG_MAX,))]
    for l, x in enumerate(chs):
        if family == "v8":                                # legacy Detect: dense 3x3 -> 3x3 -> 1x1
            E += _conv_entries(f"{p}.cv3.{l}.0", x, c3, 3)
            E += _conv_entries(f"{p}.cv3.{l}.1", c3, c3, 3)
        else:                                             # (DWConv 3x3 + Conv 1x1) x 2 -> 1x1
            E += _conv_entries(f"{p}.cv3.{l}.0.0", x, x, 3, g=x)
            E += _conv_entries(f"{p}.cv3.{l}.0.1", x, c3, 1)
            E += _conv_entries(f"{p}.cv3.{l}.1.0", c3, c3, 3, g=c3)
            E += _conv_entries(f"{p}.cv3.{l}.1.1", c3, c3, 1)
        E += [(f"{p}.cv3.{l}.2.weight", (nc, c3, 1, 1)), (f"{p}.cv3.{l}.2.bias", (nc,))]
    E += [(f"{p}.dfl.conv.weight", (1, REG_MAX, 1, 1))]
    E += _conv_entries(f"{p}.proto.cv1", chs[0], npr, 3)
    E += [(f"{p}.proto.upsample.weight", (npr, npr, 2, 2)), (f"{p}.proto.upsample.bias", (npr,))]
    E += _conv_entries(f"{p}.proto.cv2", npr, npr, 3)
    E += _conv_entries(f"{p}.proto.cv3", npr, NM, 1)
    for l, x in enumerate(chs):
        E += _conv_entries(f"{p}.cv4.{l}.0", x, c4, 3)
        E += _conv_entries(f"{p}.cv4.{l}.1", c4, c4, 3)
        E += [(f"{p}.cv4.{l}.2.weight", (NM, c4, 1, 1)), (f"{p}.cv4.{l}.2.bias", (NM,))]
    return E


# ---- NMS (utils/ops.non_max_suppression with torchvision.ops.nms restated) ---------------------------------------------
def nms_greedy(boxes: Tensor, scores: Tensor, iou_thres: float) -> Tensor:
    """torchvision.ops.nms: visit boxes by descending score, drop every later box whose IoU with a kept one is > iou_thres.
    IoU in the boxes' dtype: inter / (area_i + area_j - inter). Ties in score: lower index first (the library leaves it open)."""
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    x1, y1, x2, y2 = b.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    n = b.shape[0]
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        xx1, yy1 = torch.maximum(x1[i], x1[i + 1:]), torch.maximum(y1[i], y1[i + 1:])
        xx2, yy2 = torch.minimum(x2[i], x2[i + 1:]), torch.minimum(y2[i], y2[i + 1:])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        dead[i + 1:] |= ovr > iou_thres
    return order[torch.tensor(keep, dtype=torch.long)]


def nms_postprocess(boxes: Tensor, scores: Tensor, coeff: Optional[Tensor], conf: float, iou: float = IOU_THRES,
                    max_det: int = MAX_DET):
    """ops.non_max_suppression for one image, multi_label False, agnostic False: boxes [A,4] xyxy px, scores [A,nc], coeff [A,32].
    -> det [n,6] (xyxy, conf, cls) sorted by conf descending, anchor index [n], coeff [n,32]."""
    s, j = scores.max(1)
    cand = torch.nonzero(s > conf).squeeze(1)
    s, j, bx = s[cand], j[cand], boxes[cand]
    off = j.to(bx.dtype) * MAX_WH
    keep = nms_greedy(bx + off[:, None], s, iou)[:max_det]
    det = torch.cat((bx[keep], s[keep, None], j[keep, None].to(bx.dtype)), 1)
    return det, cand[keep], (coeff[cand[keep]] if coeff is not None else None)


class SegOracle(Oracle):
    """YOLOv8-seg / YOLO11-seg forward over an unfused state dict; numeric modes as Oracle (fp32 / bf16emu / fp64)."""

    def __init__(self, state: Dict[str, Tensor], family: str, variant: str = "n", nc: int = 80, mode: str = "fp32",
                 tap: Optional[Callable[[str, Tensor], None]] = None):
        self.family = family
        super().__init__(state, variant, nc, True, mode, tap)
        self.tbl = layer_table(family, variant)
        self.ch, self.reps = resolve(family, variant)
        self.hi, self.chs, _, _, _, _ = head_info(family, variant, nc)

    # Bottleneck with any hidden width (the weights say which)
    def c3k(self, x: Tensor, p: str, shortcut: bool) -> Tensor:
        a = self.conv(x, f"{p}.cv1")
        for t in range(2):
            a = self.bottleneck(a, f"{p}.m.{t}", shortcut)
        return self.conv(torch.cat((a, self.conv(x, f"{p}.cv2")), 1), f"{p}.cv3")

    def c3k2(self, x: Tensor, p: str, n: int, c3k: bool, shortcut: bool = True) -> Tensor:
        y = list(self.conv(x, f"{p}.cv1").chunk(2, 1))
        for j in range(n):
            y.append(self.c3k(y[-1], f"{p}.m.{j}", shortcut) if c3k else self.bottleneck(y[-1], f"{p}.m.{j}", shortcut))
        return self.conv(torch.cat(y, 1), f"{p}.cv2")

    def c2psa(self, x: Tensor, p: str, n: int) -> Tensor:
        c = x.shape[1] // 2
        a, b = self.conv(x, f"{p}.cv1").split((c, c), 1)
        for j in range(n):
            q = f"{p}.m.{j}"
            b = self.attention(b, f"{q}.attn")
            f = self.conv(b, f"{q}.ffn.0")
            b = self.conv(f, f"{q}.ffn.1", act=False, res=b)
        return self.conv(torch.cat((a, b), 1), f"{p}.cv2")

    def features(self, x: Tensor):
        outs: List[Tensor] = []
        for i, e in enumerate(self.tbl):
            kind, p = e["kind"], f"model.{i}"
            if kind == "cat":
                x = torch.cat([outs[j] if j >= 0 else outs[i + j] for j in e["f"]], 1)
            elif kind == "up":
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            elif kind == "conv":
                x = self.conv(x, p, s=e["s"])
            elif kind == "c2f":
                x = self.c2f(x, p, self.reps[i], e["shortcut"], cib=False)
            elif kind == "c3k2":
                x = self.c3k2(x, p, self.reps[i], e["c3k"])
            elif kind == "sppf":
                x = self.sppf(x, p)
            elif kind == "c2psa":
                x = self.c2psa(x, p, self.reps[i])
            outs.append(x)
        idx = (15, 18, 21) if self.family == "v8" else (16, 19, 22)
        return tuple(outs[i] for i in idx)

    def head_raw(self, feats):
        p = f"model.{self.hi}"
        boxes, clss, cfs = [], [], []
        for l, x in enumerate(feats):
            B = x.shape[0]
            b = self.conv(x, f"{p}.cv2.{l}.0")
            b = self.conv(b, f"{p}.cv2.{l}.1")
            b = self.conv(b, f"{p}.cv2.{l}.2", act=False, keep_fp32=True)
            if self.family == "v8":
                c = self.conv(x, f"{p}.cv3.{l}.0")
                c = self.conv(c, f"{p}.cv3.{l}.1")
            else:
                c = self.conv(x, f"{p}.cv3.{l}.0.0", g=x.shape[1])
                c = self.conv(c, f"{p}.cv3.{l}.0.1")
                c = self.conv(c, f"{p}.cv3.{l}.1.0", g=c.shape[1])
                c = self.conv(c, f"{p}.cv3.{l}.1.1")
            c = self.conv(c, f"{p}.cv3.{l}.2", act=False, keep_fp32=True)
            m = self.conv(x, f"{p}.cv4.{l}.0")
            m = self.conv(m, f"{p}.cv4.{l}.1")
            m = self.conv(m, f"{p}.cv4.{l}.2", act=False, keep_fp32=True)
            boxes.append(b.reshape(B, 4 * REG_MAX, -1))
            clss.append(c.reshape(B, self.nc, -1))
            cfs.append(m.reshape(B, NM, -1))
        y = self.conv(feats[0], f"{p}.proto.cv1")
        w, bb = self.w[f"{p}.proto.upsample"]
        y = self._t(f"{p}.proto.upsample", self.q(F.conv_transpose2d(y, w, bb, stride=2)))
        y = self.conv(y, f"{p}.proto.cv2")
        proto = self.conv(y, f"{p}.proto.cv3")
        return torch.cat(boxes, 2), torch.cat(clss, 2), torch.cat(cfs, 2), proto

    def forward(self, im_u8_bgr_nhwc: Tensor, conf: float = 0.25, iou: float = IOU_THRES) -> dict:
        """letterboxed uint8 batch -> dict(det: list of [n,6], idx: list of [n], coeff: list of [n,32], proto [B,32,Hp,Wp], boxes, scores).
        Detect._inference decodes to xywh and NMS converts back to xyxy; the round trip (cx -+ w/2) is kept, it is part of the
        reference's arithmetic."""
        x = self.preprocess(im_u8_bgr_nhwc)
        feats = self.features(x)
        bl, cl, cf, proto = self.head_raw(feats)
        shapes = [tuple(f.shape[-2:]) for f in feats]
        boxes, scores = self.decode(bl, cl, shapes)                    # xyxy
        cxy = (boxes[..., :2] + boxes[..., 2:]) / 2                    # dist2bbox(xywh=True): c = (x1y1 + x2y2)/2, wh = x2y2 - x1y1
        wh = boxes[..., 2:] - boxes[..., :2]
        boxes = torch.cat((cxy - wh / 2, cxy + wh / 2), -1)            # ops.xywh2xyxy
        dets, idxs, cfs = [], [], []
        for b in range(boxes.shape[0]):
            d, i, c = nms_postprocess(boxes[b].float(), scores[b].float(), cf[b].permute(1, 0).float(), conf, iou)
            dets.append(d); idxs.append(i); cfs.append(c)
        return dict(det=dets, idx=idxs, coeff=cfs, proto=proto.float(), boxes=boxes.float(), scores=scores.float(), feats=feats)


def count_params(family: str, variant: str, nc: int = 80) -> int:
    n = 0
    for name, shp in expected_state(family, variant, nc):
        if name.endswith("running_mean") or name.endswith("running_var"):
            continue
        n += int(math.prod(shp))
    return n
