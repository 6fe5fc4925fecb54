"""YOLOv10 architecture arithmetic used by the host side (weight naming/shapes, variant detection).

The executable graph itself is built natively by the engine (csrc/engine.hip: build_graph) from the same scale table;
this module only resolves channel widths / repeats per variant. Spec: SURVEY.md Appendix A.1/A.3/A.4 [U]
(the `ultralytics` yaml + parse_model rules behind `YOLO(path)`, reference yolo_seg/app.py:45).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

# variant -> (depth, width, max_channels)
SCALES: Dict[str, Tuple[float, float, int]] = {
    "n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
    "b": (0.67, 1.00, 512), "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512),
}
REG_MAX = 16
NM = 32
MAX_DET = 300

# (kind, yaml c2, yaml repeats) per backbone/neck index; variant-dependent block choice resolved below
_BASE = [
    ("conv", 64, 1), ("conv", 128, 1), ("c2f", 128, 3), ("conv", 256, 1), ("c2f", 256, 6),
    ("scdown", 512, 1), ("L6", 512, 6), ("scdown", 1024, 1), ("L8", 1024, 3), ("sppf", 1024, 1),
    ("psa", 1024, 1), ("up", 0, 1), ("cat", 6, 1), ("L13", 512, 3), ("up", 0, 1), ("cat", 4, 1),
    ("c2f", 256, 3), ("conv", 256, 1), ("cat", 13, 1), ("L19", 512, 3), ("scdown", 512, 1),
    ("cat", 10, 1), ("L22", 1024, 3),
]


def _block_choice(variant: str, tag: str) -> Tuple[str, bool, bool]:
    """-> (kind, shortcut, lk) for the variant-dependent layers."""
    v = variant
    if tag == "L6":
        return ("c2fcib", True, False) if v == "x" else ("c2f", True, False)
    if tag == "L8":
        if v == "n":
            return ("c2f", True, False)
        return ("c2fcib", True, v == "s")
    if tag == "L13":
        return ("c2f", False, False) if v in "nsm" else ("c2fcib", True, False)
    if tag == "L19":
        return ("c2f", False, False) if v in "ns" else ("c2fcib", True, False)
    if tag == "L22":
        return ("c2fcib", True, v in "ns")
    raise KeyError(tag)


def make_divisible(x: float, d: int = 8) -> int:
    return int(math.ceil(x / d) * d)


def layer_plan(variant: str) -> List[dict]:
    """Resolved per-layer records for indices 0..22: kind, name, c1, c2 and block parameters."""
    depth, width, maxc = SCALES[variant]
    plan: List[dict] = []
    ch: List[int] = []
    for i, (kind, c2y, ny) in enumerate(_BASE):
        rec = dict(i=i, name=f"model.{i}", kind=kind)
        c1 = 3 if i == 0 else ch[-1]
        if kind == "up":
            c2 = c1
        elif kind == "cat":
            rec["src"] = c2y
            c2 = c1 + ch[c2y]
        else:
            c2 = make_divisible(min(c2y, maxc) * width, 8)
        n = max(round(ny * depth), 1) if ny > 1 else ny
        shortcut, lk = False, False
        if kind.startswith("L"):
            kind, shortcut, lk = _block_choice(variant, kind)
        elif kind == "c2f":
            shortcut = i in (2, 4)
        rec.update(kind=kind, c1=c1, c2=c2, n=n, shortcut=shortcut, lk=lk)
        if kind in ("c2f", "c2fcib"):
            rec["c"] = int(c2 * 0.5)
        if kind in ("conv", "scdown"):
            rec.update(k=3, s=2)
        if kind == "psa":
            c = int(c1 * 0.5)
            nh = c // 64
            hd = c // nh
            rec.update(c=c, nh=nh, hd=hd, kd=int(hd * 0.5))
        plan.append(rec)
        ch.append(c2)
    return plan


def resolve_channels(variant: str) -> Tuple[List[int], List[int]]:
    p = layer_plan(variant)
    return [r["c2"] for r in p], [r["n"] for r in p]


def head_dims(variant: str, nc: int) -> Tuple[Tuple[int, int, int], int, int, int]:
    """v10Detect (A.4): ((ch P3,P4,P5), c2 box hidden, c3 cls hidden, c4 coeff hidden)."""
    ch, _ = resolve_channels(variant)
    chs = (ch[16], ch[19], ch[22])
    return chs, max(16, chs[0] // 4, REG_MAX * 4), max(chs[0], min(nc, 100)), max(chs[0] // 4, NM)
