"""CPU ORACLE (test infrastructure, NOT product code) for the YOLOv10 predict hot path.

PARITY UNPINNED: the arithmetic of the reference's hot path lives in the un-pinned, un-vendored
third-party package `ultralytics` (reference pyproject.toml:23; imported at yolo_seg/app.py:7,
yolo_seg/yolo_with_deva.py:12). It is absent from /root/reference, not installed, and the reference holds
no tests, golden vectors or fixtures for this path. This file is therefore a
restatement of the *published* ultralytics / THU-MIG YOLOv10 algorithm as specified in SURVEY.md
Appendix A ([U]), anchored on the reference's own call sites:
    YOLO(path)                       yolo_seg/app.py:45, yolo_seg/yolo_with_deva.py:226
    .predict(source, conf, retina_masks, device)
                                     yolo_seg/app.py:49,91 ; yolo_seg/yolo_with_deva.py:51 ;
                                     dev_tools/auto_speed_calc.py:62 ; dev_tools/classify/cls_bbox_dataset_generate.py:48
Its structural correctness is gated by the parameter / FLOP identities of SURVEY.md Appendix B against the
table the reference publishes at README.md:48-53 (tests/test_oracle_structure.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. The product
package (yolo-puncture_amd/) never does.

Everything here is plain functional PyTorch on CPU: F.conv2d & friends, no nn.Module zoo.

Two numeric modes:
  * "fp32"    - what the reference's PyTorch CPU path computes.
  * "bf16emu" - every tensor that the engine materialises in HBM (weights and each fused op's output) is
                rounded to bfloat16; all arithmetic inside one fused op is fp32. This is the checker for the
                engine's bf16 mode: differences against it come from accumulation order only.
The fusion boundaries of bf16emu (where a rounding happens) are part of this spec and are stated per op.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------------------------
# A.1 conventions  [U]
# --------------------------------------------------------------------------------------------------------
# [depth, width, max_channels] per variant  
SCALES: Dict[str, Tuple[float, float, int]] = {
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
    "b": (0.67, 1.00, 512),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.25, 512),
}
BN_EPS = 1e-3
REG_MAX = 16
MAX_DET = 300
NM = 32  # mask coefficients


def make_divisible(x: float, d: int = 8) -> int:
    return int(math.ceil(x / d) * d)


def scaled_c(c: int, variant: str) -> int:
    _, width, maxc = SCALES[variant]
    return make_divisible(min(c, maxc) * width, 8)


def scaled_n(n: int, variant: str) -> int:
    depth = SCALES[variant][0]
    return max(round(n * depth), 1) if n > 1 else n


# Layer table of SURVEY Appendix A.3. Entries: (from, kind, yaml_c2, yaml_repeats, extra)
# kind in {conv, c2f, c2fcib, scdown, sppf, psa, up, cat}; extra = dict(shortcut=, lk=, k=, s=)
def layer_table(variant: str) -> List[dict]:
    v = variant
    L: List[dict] = []

    def add(frm, kind, c2=0, n=1, **kw):
        L.append(dict(f=frm, kind=kind, c2=c2, n=n, **kw))

    add(-1, "conv", 64, k=3, s=2)                      # 0  P1/2
    add(-1, "conv", 128, k=3, s=2)                     # 1  P2/4
    add(-1, "c2f", 128, 3, shortcut=True)              # 2
    add(-1, "conv", 256, k=3, s=2)                     # 3  P3/8
    add(-1, "c2f", 256, 6, shortcut=True)              # 4
    add(-1, "scdown", 512, k=3, s=2)                   # 5  P4/16
    if v == "x":
        add(-1, "c2fcib", 512, 6, shortcut=True, lk=False)  # 6
    else:
        add(-1, "c2f", 512, 6, shortcut=True)          # 6
    add(-1, "scdown", 1024, k=3, s=2)                  # 7  P5/32
    if v == "n":
        add(-1, "c2f", 1024, 3, shortcut=True)         # 8
    elif v == "s":
        add(-1, "c2fcib", 1024, 3, shortcut=True, lk=True)
    else:
        add(-1, "c2fcib", 1024, 3, shortcut=True, lk=False)
    add(-1, "sppf", 1024, k=5)                         # 9
    add(-1, "psa", 1024)                               # 10
    add(-1, "up")                                      # 11
    add([-1, 6], "cat")                                # 12
    if v in ("n", "s", "m"):
        add(-1, "c2f", 512, 3, shortcut=False)         # 13
    else:
        add(-1, "c2fcib", 512, 3, shortcut=True, lk=False)
    add(-1, "up")                                      # 14
    add([-1, 4], "cat")                                # 15
    add(-1, "c2f", 256, 3, shortcut=False)             # 16  P3 out
    add(-1, "conv", 256, k=3, s=2)                     # 17
    add([-1, 13], "cat")                               # 18
    if v in ("n", "s"):
        add(-1, "c2f", 512, 3, shortcut=False)         # 19  P4 out
    else:
        add(-1, "c2fcib", 512, 3, shortcut=True, lk=False)
    add(-1, "scdown", 512, k=3, s=2)                   # 20
    add([-1, 10], "cat")                               # 21
    if v in ("n", "s"):
        add(-1, "c2fcib", 1024, 3, shortcut=True, lk=True)   # 22  P5 out
    else:
        add(-1, "c2fcib", 1024, 3, shortcut=True, lk=False)
    return L


def resolve_channels(variant: str) -> Tuple[List[int], List[int]]:
    """-> (out channels per layer index 0..22, repeats per layer)."""
    tbl = layer_table(variant)
    ch: List[int] = []
    reps: List[int] = []
    for i, e in enumerate(tbl):
        if e["kind"] == "up":
            c = ch[-1]
        elif e["kind"] == "cat":
            c = sum(ch[j] if j >= 0 else ch[i + j] for j in e["f"])
        else:
            c = scaled_c(e["c2"], variant)
        ch.append(c)
        reps.append(scaled_n(e["n"], variant))
    return ch, reps


def head_dims(variant: str, nc: int) -> Tuple[Tuple[int, int, int], int, int, int]:
    """v10Detect dims (A.4): (ch_P3,P4,P5), c2(box hidden), c3(cls hidden), c4(coeff hidden)."""
    ch, _ = resolve_channels(variant)
    chs = (ch[16], ch[19], ch[22])
    c2 = max(16, chs[0] // 4, REG_MAX * 4)
    c3 = max(chs[0], min(nc, 100))
    c4 = max(chs[0] // 4, NM)
    return chs, c2, c3, c4


# --------------------------------------------------------------------------------------------------------
# Expected state-dict layout (A.8): names + shapes of an *unfused* ultralytics checkpoint
# --------------------------------------------------------------------------------------------------------
def _conv_entries(name: str, c1: int, c2: int, k: int, g: int = 1) -> List[Tuple[str, Tuple[int, ...]]]:
    return [
        (f"{name}.conv.weight", (c2, c1 // g, k, k)),
        (f"{name}.bn.weight", (c2,)),
        (f"{name}.bn.bias", (c2,)),
        (f"{name}.bn.running_mean", (c2,)),
        (f"{name}.bn.running_var", (c2,)),
    ]


def _cib_entries(p: str, c: int, lk: bool) -> List[Tuple[str, Tuple[int, ...]]]:
    # CIB(c1=c, c2=c, e=1.0): c_ = c ; cv1 = [dw3(c), pw(c->2c), dw3/RepVGGDW(2c), pw(2c->c), dw3(c)]
    E: List[Tuple[str, Tuple[int, ...]]] = []
    E += _conv_entries(f"{p}.cv1.0", c, c, 3, g=c)
    E += _conv_entries(f"{p}.cv1.1", c, 2 * c, 1)
    if lk:
        E += _conv_entries(f"{p}.cv1.2.conv", 2 * c, 2 * c, 7, g=2 * c)
        E += _conv_entries(f"{p}.cv1.2.conv1", 2 * c, 2 * c, 3, g=2 * c)
    else:
        E += _conv_entries(f"{p}.cv1.2", 2 * c, 2 * c, 3, g=2 * c)
    E += _conv_entries(f"{p}.cv1.3", 2 * c, c, 1)
    E += _conv_entries(f"{p}.cv1.4", c, c, 3, g=c)
    return E


def expected_state(variant: str, nc: int = 80, seg: bool = False,
                   one2many: bool = False) -> List[Tuple[str, Tuple[int, ...]]]:
    """Names/shapes of the unfused parameters+buffers the oracle consumes (A.8).

    one2many=True additionally lists the training-only `cv2`/`cv3` twins and `dfl.conv.weight`
    so the A.8 totals (N 2,775,520; S 8,128,272; X 31,808,960) can be asserted.
    """
    tbl = layer_table(variant)
    ch, reps = resolve_channels(variant)
    E: List[Tuple[str, Tuple[int, ...]]] = []
    for i, e in enumerate(tbl):
        p = f"model.{i}"
        kind = e["kind"]
        if kind in ("up", "cat"):
            continue
        c1 = 3 if i == 0 else (ch[i - 1] if isinstance(e["f"], int) else None)
        c2 = ch[i]
        if kind == "conv":
            E += _conv_entries(p, c1, c2, e["k"])
        elif kind in ("c2f", "c2fcib"):
            c = int(c2 * 0.5)
            n = reps[i]
            E += _conv_entries(f"{p}.cv1", c1, 2 * c, 1)
            E += _conv_entries(f"{p}.cv2", (2 + n) * c, c2, 1)
            for j in range(n):
                if kind == "c2f":
                    E += _conv_entries(f"{p}.m.{j}.cv1", c, c, 3)
                    E += _conv_entries(f"{p}.m.{j}.cv2", c, c, 3)
                else:
                    E += _cib_entries(f"{p}.m.{j}", c, e["lk"])
        elif kind == "scdown":
            E += _conv_entries(f"{p}.cv1", c1, c2, 1)
            E += _conv_entries(f"{p}.cv2", c2, c2, e["k"], g=c2)
        elif kind == "sppf":
            c_ = c1 // 2
            E += _conv_entries(f"{p}.cv1", c1, c_, 1)
            E += _conv_entries(f"{p}.cv2", 4 * c_, c2, 1)
        elif kind == "psa":
            c = int(c1 * 0.5)
            nh = c // 64
            hd = c // nh
            kd = int(hd * 0.5)
            E += _conv_entries(f"{p}.cv1", c1, 2 * c, 1)
            E += _conv_entries(f"{p}.cv2", 2 * c, c1, 1)
            E += _conv_entries(f"{p}.attn.qkv", c, c + 2 * kd * nh, 1)
            E += _conv_entries(f"{p}.attn.proj", c, c, 1)
            E += _conv_entries(f"{p}.attn.pe", c, c, 3, g=c)
            E += _conv_entries(f"{p}.ffn.0", c, 2 * c, 1)
            E += _conv_entries(f"{p}.ffn.1", 2 * c, c, 1)
    # head (index 23)
    chs, c2, c3, c4 = head_dims(variant, nc)
    p = "model.23"
    box_names = ["one2one_cv2"] + (["cv2"] if one2many else [])
    cls_names = ["one2one_cv3"] + (["cv3"] if one2many else [])
    for bn in box_names:
        for l, x in enumerate(chs):
            E += _conv_entries(f"{p}.{bn}.{l}.0", x, c2, 3)
            E += _conv_entries(f"{p}.{bn}.{l}.1", c2, c2, 3)
            E += [(f"{p}.{bn}.{l}.2.weight", (4 * REG_MAX, c2, 1, 1)), (f"{p}.{bn}.{l}.2.bias", (4 * REG_MAX,))]
    for cn in cls_names:
        for l, x in enumerate(chs):
            E += _conv_entries(f"{p}.{cn}.{l}.0.0", x, x, 3, g=x)
            E += _conv_entries(f"{p}.{cn}.{l}.0.1", x, c3, 1)
            E += _conv_entries(f"{p}.{cn}.{l}.1.0", c3, c3, 3, g=c3)
            E += _conv_entries(f"{p}.{cn}.{l}.1.1", c3, c3, 1)
            E += [(f"{p}.{cn}.{l}.2.weight", (nc, c3, 1, 1)), (f"{p}.{cn}.{l}.2.bias", (nc,))]
    if one2many:
        E += [(f"{p}.dfl.conv.weight", (1, REG_MAX, 1, 1))]
    if seg:
        # Build-defined "v10-seg" head: Proto on the P3 output + cv4 coefficient branches.
        npr = chs[0]  # Proto(c1=ch[0], c_=npr, c2=nm); Segment uses npr=256*width == ch[0] for n/s/m (see A.7)
        E += _conv_entries(f"{p}.proto.cv1", chs[0], npr, 3)
        E += [(f"{p}.proto.upsample.weight", (npr, npr, 2, 2)), (f"{p}.proto.upsample.bias", (npr,))]
        E += _conv_entries(f"{p}.proto.cv2", npr, npr, 3)
        E += _conv_entries(f"{p}.proto.cv3", npr, NM, 1)
        for l, x in enumerate(chs):
            E += _conv_entries(f"{p}.cv4.{l}.0", x, c4, 3)
            E += _conv_entries(f"{p}.cv4.{l}.1", c4, c4, 3)
            E += [(f"{p}.cv4.{l}.2.weight", (NM, c4, 1, 1)), (f"{p}.cv4.{l}.2.bias", (NM,))]
    return E


def count_params(entries: List[Tuple[str, Tuple[int, ...]]], with_bn_stats: bool = False) -> int:
    """Learnable parameter count (conv weights, BN gamma/beta, biased Conv2d, dfl) as ultralytics counts it."""
    n = 0
    for name, shp in entries:
        if not with_bn_stats and (name.endswith("running_mean") or name.endswith("running_var")):
            continue
        n += int(math.prod(shp))
    return n


# --------------------------------------------------------------------------------------------------------
# numeric mode helpers
# --------------------------------------------------------------------------------------------------------
def _q_bf16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def _q_id(x: Tensor) -> Tensor:
    return x


class Oracle:
    """Functional YOLOv10 forward over a state dict (fp32 tensors, ultralytics key names, unfused BN).

    fuse(): A.1 'Inference always runs fused' - W' = W*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps);
            RepVGGDW merged into one 7x7 depthwise (A.2).
    """

    def __init__(self, state: Dict[str, Tensor], variant: str = "s", nc: int = 80, seg: bool = False,
                 mode: str = "fp32", tap: Optional[Callable[[str, Tensor], None]] = None):
        assert mode in ("fp32", "bf16emu", "fp64")
        self.variant, self.nc, self.seg, self.mode = variant, nc, seg, mode
        self.q = _q_bf16 if mode == "bf16emu" else _q_id
        self.dt = torch.float64 if mode == "fp64" else torch.float32
        self.tap = tap
        self.tbl = layer_table(variant)
        self.ch, self.reps = resolve_channels(variant)
        self.w: Dict[str, Tuple[Tensor, Tensor]] = {}
        self._fuse(state)

    # ---- load-time folding ------------------------------------------------------------------------------
    def _fold(self, st: Dict[str, Tensor], name: str) -> Tuple[Tensor, Tensor]:
        w = st[f"{name}.conv.weight"].to(torch.float32)
        g, b = st[f"{name}.bn.weight"].float(), st[f"{name}.bn.bias"].float()
        m, v = st[f"{name}.bn.running_mean"].float(), st[f"{name}.bn.running_var"].float()
        scale = g / torch.sqrt(v + BN_EPS)
        return w * scale.view(-1, 1, 1, 1), b - m * scale

    def _fuse(self, st: Dict[str, Tensor]) -> None:
        names = set()
        for k in st:
            if k.endswith(".conv.weight") and not k.endswith("dfl.conv.weight"):
                names.add(k[: -len(".conv.weight")])
        folded: Dict[str, Tuple[Tensor, Tensor]] = {n: self._fold(st, n) for n in names}
        # RepVGGDW: '<p>.conv' (7x7) + '<p>.conv1' (3x3) -> one 7x7 at '<p>'
        for n in list(folded):
            if n.endswith(".conv1") and (n[:-1] in folded):
                w7, b7 = folded[n[:-1]]
                w3, b3 = folded[n]
                folded[n[: -len(".conv1")]] = (w7 + F.pad(w3, [2, 2, 2, 2]), b7 + b3)
                del folded[n], folded[n[:-1]]
        # plain biased Conv2d / ConvTranspose2d
        for k in st:
            if k.endswith(".weight") and (k[:-7] + ".bias") in st and not k.endswith(".bn.weight"):
                folded[k[:-7]] = (st[k].float(), st[k[:-7] + ".bias"].float())
        for n, (w, b) in folded.items():
            self.w[n] = (self.q(w).to(self.dt), b.to(self.dt))   # weights rounded in bf16emu; bias stays fp32

    # ---- fused ops (each returns a tensor rounded by self.q == one HBM materialisation in the engine) -----
    def _t(self, name: str, x: Tensor) -> Tensor:
        if self.tap is not None:
            self.tap(name, x)
        return x

    def conv(self, x: Tensor, name: str, s: int = 1, act: bool = True, g: int = 1,
             res: Optional[Tensor] = None, keep_fp32: bool = False) -> Tensor:
        """Conv(+folded BN)(+SiLU)(+residual added AFTER the activation) - A.2 `Conv`, `Bottleneck`.
        keep_fp32: the output is not rounded in bf16emu (the engine stores the head's final logits as fp32)."""
        w, b = self.w[name]
        k = w.shape[-1]
        y = F.conv2d(x, w, b, stride=s, padding=k // 2, groups=g)
        if act:
            y = F.silu(y)
        if res is not None:
            y = y + res
        return self._t(name, y if keep_fp32 else self.q(y))

    def bottleneck(self, x: Tensor, p: str, shortcut: bool) -> Tensor:
        y = self.conv(x, f"{p}.cv1")
        return self.conv(y, f"{p}.cv2", res=x if shortcut else None)

    def cib(self, x: Tensor, p: str, shortcut: bool, lk: bool) -> Tensor:
        c = x.shape[1]
        y = self.conv(x, f"{p}.cv1.0", g=c)
        y = self.conv(y, f"{p}.cv1.1")
        # RepVGGDW forward = SiLU(conv7(x)+conv3(x)) == merged 7x7 + SiLU ; else Conv(dw3)+SiLU
        y = self.conv(y, f"{p}.cv1.2", g=2 * c)
        y = self.conv(y, f"{p}.cv1.3")
        return self.conv(y, f"{p}.cv1.4", g=c, res=x if shortcut else None)

    def c2f(self, x: Tensor, p: str, n: int, shortcut: bool, cib: bool, lk: bool = False) -> Tensor:
        y = list(self.conv(x, f"{p}.cv1").chunk(2, 1))
        for j in range(n):
            y.append(self.cib(y[-1], f"{p}.m.{j}", shortcut, lk) if cib
                     else self.bottleneck(y[-1], f"{p}.m.{j}", shortcut))
        return self.conv(torch.cat(y, 1), f"{p}.cv2")

    def scdown(self, x: Tensor, p: str, s: int) -> Tensor:
        y = self.conv(x, f"{p}.cv1")
        return self.conv(y, f"{p}.cv2", s=s, act=False, g=y.shape[1])

    def sppf(self, x: Tensor, p: str) -> Tensor:
        y = [self.conv(x, f"{p}.cv1")]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.conv(torch.cat(y, 1), f"{p}.cv2")

    def attention(self, x: Tensor, p: str) -> Tensor:
        """A.2 `Attention`. bf16emu roundings: qkv output; softmax probabilities P; attention output o;
        (o + pe(v)) is ONE fused op (pe depthwise conv epilogue adds o); proj(+residual) is one fused op."""
        B, C, H, W = x.shape
        nh = C // 64
        hd = C // nh
        kd = int(hd * 0.5)
        N = H * W
        qkv = self.conv(x, f"{p}.qkv", act=False)
        q, k, v = qkv.view(B, nh, 2 * kd + hd, N).split([kd, kd, hd], dim=2)
        attn = (q.transpose(-2, -1) @ k) * (kd ** -0.5)
        attn = self.q(attn.softmax(dim=-1))
        o = self._t(f"{p}.o", self.q((v @ attn.transpose(-2, -1)).reshape(B, C, H, W)))
        y = self.conv(v.reshape(B, C, H, W), f"{p}.pe", act=False, g=C, res=o)
        return self.conv(y, f"{p}.proj", act=False, res=x)       # PSA: b = b + attn(b)

    def psa(self, x: Tensor, p: str) -> Tensor:
        c = x.shape[1] // 2
        a, b = self.conv(x, f"{p}.cv1").split((c, c), 1)
        b = self.attention(b, f"{p}.attn")
        f = self.conv(b, f"{p}.ffn.0")
        b = self.conv(f, f"{p}.ffn.1", act=False, res=b)           # b = b + ffn(b)
        return self.conv(torch.cat((a, b), 1), f"{p}.cv2")

    # ---- preprocessing of an already-letterboxed batch (A.5 step 3) ------------------------------------------
    def preprocess(self, im_u8_bgr_nhwc: Tensor) -> Tensor:
        """uint8 [B,H,W,3] BGR -> float [B,3,H,W] RGB / 255. bf16emu: input rounded to bf16 after the division."""
        x = im_u8_bgr_nhwc.flip(-1).permute(0, 3, 1, 2).contiguous().to(torch.float32) / 255.0
        return self.q(x).to(self.dt)

    # ---- backbone + neck (A.3) --------------------------------------------------------------------------------
    def features(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        outs: List[Tensor] = []
        for i, e in enumerate(self.tbl):
            kind = e["kind"]
            p = f"model.{i}"
            if kind == "cat":
                x = torch.cat([outs[j] if j >= 0 else outs[i + j] for j in e["f"]], 1)
            elif kind == "up":
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            elif kind == "conv":
                x = self.conv(x, p, s=e["s"])
            elif kind == "c2f":
                x = self.c2f(x, p, self.reps[i], e["shortcut"], cib=False)
            elif kind == "c2fcib":
                x = self.c2f(x, p, self.reps[i], e["shortcut"], cib=True, lk=e["lk"])
            elif kind == "scdown":
                x = self.scdown(x, p, e["s"])
            elif kind == "sppf":
                x = self.sppf(x, p)
            elif kind == "psa":
                x = self.psa(x, p)
            outs.append(x)
        return outs[16], outs[19], outs[22]

    # ---- v10Detect one-to-one head (A.4) ----------------------------------------------------------------------
    def head_raw(self, feats) -> Tuple[Tensor, Tensor, Optional[Tensor], Optional[Tensor]]:
        """-> box logits [B,64,A], cls logits [B,nc,A], coeff [B,32,A]|None, proto [B,32,Hp,Wp]|None.
        Anchor order: P3,P4,P5; row-major within a level."""
        p = "model.23"
        boxes, clss, cfs = [], [], []
        for l, x in enumerate(feats):
            B = x.shape[0]
            b = self.conv(x, f"{p}.one2one_cv2.{l}.0")
            b = self.conv(b, f"{p}.one2one_cv2.{l}.1")
            b = self.conv(b, f"{p}.one2one_cv2.{l}.2", act=False, keep_fp32=True)
            c = self.conv(x, f"{p}.one2one_cv3.{l}.0.0", g=x.shape[1])
            c = self.conv(c, f"{p}.one2one_cv3.{l}.0.1")
            c = self.conv(c, f"{p}.one2one_cv3.{l}.1.0", g=c.shape[1])
            c = self.conv(c, f"{p}.one2one_cv3.{l}.1.1")
            c = self.conv(c, f"{p}.one2one_cv3.{l}.2", act=False, keep_fp32=True)
            boxes.append(b.reshape(B, 4 * REG_MAX, -1))
            clss.append(c.reshape(B, self.nc, -1))
            if self.seg:
                m = self.conv(x, f"{p}.cv4.{l}.0")
                m = self.conv(m, f"{p}.cv4.{l}.1")
                m = self.conv(m, f"{p}.cv4.{l}.2", act=False, keep_fp32=True)
                cfs.append(m.reshape(B, NM, -1))
        proto = None
        if self.seg:
            x = feats[0]
            y = self.conv(x, f"{p}.proto.cv1")
            w, b = self.w[f"{p}.proto.upsample"]
            y = self._t(f"{p}.proto.upsample", self.q(F.conv_transpose2d(y, w, b, stride=2)))
            y = self.conv(y, f"{p}.proto.cv2")
            proto = self.conv(y, f"{p}.proto.cv3")
        return torch.cat(boxes, 2), torch.cat(clss, 2), (torch.cat(cfs, 2) if self.seg else None), proto

    @staticmethod
    def make_anchors(shapes: List[Tuple[int, int]], strides=(8, 16, 32), dt=torch.float32):
        """A.4 make_anchors, offset 0.5: -> anchor points [A,2] (x,y) in grid units, stride [A]."""
        pts, st = [], []
        for (h, w), s in zip(shapes, strides):
            sx = torch.arange(w, dtype=dt) + 0.5
            sy = torch.arange(h, dtype=dt) + 0.5
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            pts.append(torch.stack((xx, yy), -1).view(-1, 2))
            st.append(torch.full((h * w,), float(s), dtype=dt))
        return torch.cat(pts), torch.cat(st)

    def decode(self, box_logits: Tensor, cls_logits: Tensor, shapes) -> Tuple[Tensor, Tensor]:
        """DFL expectation + dist2bbox(xyxy) * stride ; sigmoid scores (A.4). -> boxes [B,A,4], scores [B,A,nc].
        Always computed in >=fp32 (the engine's head epilogue is fp32 in both modes)."""
        B, _, A = box_logits.shape
        anc, strd = self.make_anchors(shapes, dt=self.dt)
        d = box_logits.view(B, 4, REG_MAX, A).softmax(2)
        d = (d * torch.arange(REG_MAX, dtype=self.dt).view(1, 1, REG_MAX, 1)).sum(2)   # [B,4,A] l,t,r,b
        ax, ay = anc[:, 0], anc[:, 1]
        x1, y1 = ax - d[:, 0], ay - d[:, 1]
        x2, y2 = ax + d[:, 2], ay + d[:, 3]
        boxes = torch.stack((x1, y1, x2, y2), -1) * strd.view(1, A, 1)
        return boxes, cls_logits.sigmoid().permute(0, 2, 1).contiguous()

    def forward(self, im_u8_bgr_nhwc: Tensor) -> dict:
        """Letterboxed uint8 batch -> dict(det [B,300,6], idx [B,300] anchor index, cls, coeff, proto)."""
        x = self.preprocess(im_u8_bgr_nhwc)
        feats = self.features(x)
        bl, cl, cf, proto = self.head_raw(feats)
        shapes = [tuple(f.shape[-2:]) for f in feats]
        boxes, scores = self.decode(bl, cl, shapes)
        det, idx = v10_postprocess(boxes, scores)
        out = dict(det=det.float(), idx=idx, boxes=boxes.float(), scores=scores.float(), feats=feats)
        if self.seg:
            B = det.shape[0]
            out["coeff"] = torch.gather(cf.permute(0, 2, 1), 1, idx.unsqueeze(-1).expand(B, idx.shape[1], NM)).float()
            out["proto"] = proto.float()
        return out


# --------------------------------------------------------------------------------------------------------
# A.6 detect post-process (deterministic restatement of the two-stage top-k)
# --------------------------------------------------------------------------------------------------------
def _topk_det(v: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """top-k, value descending, ties broken by LOWEST index first (torch.topk's tie order is
    unspecified, the build fixes it). v: [B,n]."""
    order = torch.sort(v, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(v, 1, order), order


def v10_postprocess(boxes: Tensor, scores: Tensor, max_det: int = MAX_DET) -> Tuple[Tensor, Tensor]:
    """boxes [B,A,4] xyxy, scores [B,A,nc] -> det [B,k,6] (x1,y1,x2,y2,score,cls), anchor idx [B,k].

    A.6 step 1: m = scores.amax(-1); idx1 = topk(m,k); gather; (s,f) = topk(scores[idx1].flatten(1), k);
    cls = f % nc ; row = f // nc. One anchor may appear several times with different classes."""
    B, A, nc = scores.shape
    k = min(max_det, A)
    m = scores.amax(-1)
    _, idx1 = _topk_det(m, k)
    b1 = torch.gather(boxes, 1, idx1.unsqueeze(-1).expand(B, k, 4))
    s1 = torch.gather(scores, 1, idx1.unsqueeze(-1).expand(B, k, nc))
    s, f = _topk_det(s1.flatten(1), k)
    cls = f % nc
    row = f // nc
    b2 = torch.gather(b1, 1, row.unsqueeze(-1).expand(B, k, 4))
    anchor = torch.gather(idx1, 1, row)
    det = torch.cat((b2, s.unsqueeze(-1), cls.to(b2.dtype).unsqueeze(-1)), -1)
    return det, anchor


def count_conv_flops(variant: str, nc: int = 80, seg: bool = False, hw: Tuple[int, int] = (640, 640)) -> float:
    """2*MAC over convolutions + PSA matmuls of the fused one-to-one graph (Appendix B definition)."""
    flops = 0.0
    H, W = hw

    def tap(name: str, y: Tensor):
        pass

    ent = dict(expected_state(variant, nc, seg))
    tbl = layer_table(variant)
    ch, reps = resolve_channels(variant)
    # spatial size per layer
    size = []
    h, w = H, W
    sizes = []
    for i, e in enumerate(tbl):
        if e["kind"] in ("conv", "scdown"):
            h, w = (h + 1) // 2, (w + 1) // 2
        elif e["kind"] == "up":
            h, w = h * 2, w * 2
        elif e["kind"] == "cat":
            pass
        sizes.append((h, w))
        if e["kind"] == "cat":
            j = e["f"][1]
            h, w = sizes[j]
            sizes[-1] = (h, w)
    lvl = {0: sizes[16], 1: sizes[19], 2: sizes[22]}
    for name, shp in ent.items():
        if not (name.endswith("conv.weight") or name.endswith(".2.weight") or name.endswith("upsample.weight")):
            continue
        if name.endswith("conv1.conv.weight"):
            continue  # RepVGGDW 3x3 is merged into the 7x7 at load
        parts = name.split(".")
        i = int(parts[1])
        if i < 23:
            oh, ow = sizes[i]
            if tbl[i]["kind"] == "scdown" and ".cv1." in name:
                oh, ow = sizes[i - 1]
        else:
            if "proto" in name:
                oh, ow = lvl[0]
                if "upsample" in name or "cv2" in name or "cv3" in name:
                    oh, ow = oh * 2, ow * 2
            else:
                oh, ow = lvl[int(parts[3])]
        c2, c1g, kh, kw = shp
        if "upsample" in name:
            # ConvTranspose2d k=2 s=2: each output pixel sees exactly one tap -> Cin*Cout MAC per output px
            flops += 2.0 * oh * ow * shp[0] * shp[1]
        else:
            flops += 2.0 * oh * ow * c2 * c1g * kh * kw
    # PSA matmuls
    c = int(ch[10] * 0.5)
    nh = c // 64
    hd = c // nh
    kd = int(hd * 0.5)
    N = sizes[10][0] * sizes[10][1]
    flops += 2.0 * nh * N * N * kd + 2.0 * nh * N * N * hd
    return flops
