"""Weights for the engine: ultralytics `.pt` reader (without ultralytics), BN folding, synthetic weights.

Replaces the loading half of `YOLO(path)` (reference yolo_seg/app.py:45, yolo_seg/yolo_with_deva.py:226).
Layout facts are SURVEY.md Appendix A.8 [U].
"""
from __future__ import annotations

import io
import math
import os
import pickle
import types
import zlib
from typing import Dict, List, Optional, Tuple

import torch

from .arch import REG_MAX, NM, head_dims, layer_plan, resolve_channels

BN_EPS = 1e-3


# --------------------------------------------------------------------------------------------------------
# ultralytics .pt reader
# --------------------------------------------------------------------------------------------------------
class _Stub:
    """Inert stand-in for any class the checkpoint pickles by reference (ultralytics.*, its nn.Modules ...).
    It only has to survive unpickling: torch restores tensors itself through persistent ids."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            self.__dict__.update(state[0] or {})
            self.__dict__.update(state[1])

    def __call__(self, *a, **k):
        return None


_STUB_CACHE: Dict[Tuple[str, str], type] = {}


def _stub_class(module: str, name: str) -> type:
    key = (module, name)
    if key not in _STUB_CACHE:
        _STUB_CACHE[key] = type(name, (_Stub,), {"__module__": module})
    return _STUB_CACHE[key]


class _StubUnpickler(pickle.Unpickler):
    _ALLOW_PREFIX = ("torch", "collections", "numpy", "builtins", "_codecs", "copyreg", "pathlib", "__builtin__")

    def find_class(self, module, name):
        if module.split(".")[0] in self._ALLOW_PREFIX and not module.startswith("torch.nn.modules"):
            return super().find_class(module, name)
        return _stub_class(module, name)


_stub_pickle = types.ModuleType("yp_stub_pickle")
_stub_pickle.Unpickler = _StubUnpickler
_stub_pickle.load = lambda f, **kw: _StubUnpickler(f, **kw).load()
_stub_pickle.__name__ = "pickle"
for _n in ("dump", "dumps", "loads", "Pickler", "HIGHEST_PROTOCOL", "PicklingError", "UnpicklingError"):
    setattr(_stub_pickle, _n, getattr(pickle, _n))


def _walk_module(obj, prefix: str, out: Dict[str, torch.Tensor]) -> None:
    d = getattr(obj, "__dict__", {})
    for kind in ("_parameters", "_buffers"):
        for k, v in (d.get(kind) or {}).items():
            if v is not None and isinstance(v, torch.Tensor):
                out[f"{prefix}{k}"] = v.detach()
    for k, sub in (d.get("_modules") or {}).items():
        if sub is not None:
            _walk_module(sub, f"{prefix}{k}.", out)


def read_ultralytics_pt(path: str) -> Tuple[Dict[str, torch.Tensor], dict]:
    """-> (state dict name->fp32 tensor, meta). meta: yaml (dict|None), names, task guess, variant guess.
    Only ever applied to a local file the user supplies; nothing is fetched."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    ckpt = torch.load(path, map_location="cpu", pickle_module=_stub_pickle, weights_only=False)
    meta: dict = {}
    if isinstance(ckpt, dict) and ("model" in ckpt or "ema" in ckpt):
        mod = ckpt.get("ema") or ckpt.get("model")
    else:
        mod = ckpt
    state: Dict[str, torch.Tensor] = {}
    if isinstance(mod, dict):              # a plain state_dict checkpoint
        state = {k: v for k, v in mod.items() if isinstance(v, torch.Tensor)}
    else:
        _walk_module(mod, "", state)
        meta["yaml"] = getattr(mod, "yaml", None)
        meta["names"] = getattr(mod, "names", None)
    state = {k: v.float() for k, v in state.items() if v.is_floating_point()}
    meta["family"] = guess_family(state)
    meta["seg"] = any(".proto." in k for k in state)
    if meta["family"] == "v10":
        meta["variant"] = guess_variant(state)
        nck = [k for k in state if k.endswith("one2one_cv3.0.2.weight")]
    else:
        meta["variant"] = guess_variant_family(state, meta["family"]) if meta["family"] else None
        hi = {"v8": 22, "11": 23}.get(meta["family"], -1)
        nck = [k for k in state if k == f"model.{hi}.cv3.0.2.weight"]
    meta["nc"] = int(state[nck[0]].shape[0]) if nck else None
    return state, meta


def guess_family(state: Dict[str, torch.Tensor]) -> Optional[str]:
    """Which ultralytics yaml a checkpoint follows: 'v10' (one-to-one head), '11' (C3k2 / C2PSA, head at model.23) or 'v8' (C2f,
    head at model.22) - the latter two are what the reference's UI offers (yolo_seg/app.py:218-223)."""
    if any(".one2one_cv2." in k for k in state):
        return "v10"
    if "model.23.cv2.0.0.conv.weight" in state and any(k.startswith("model.10.m.0.attn.") for k in state):
        return "11"
    if "model.22.cv2.0.0.conv.weight" in state and "model.9.cv2.conv.weight" in state:
        return "v8"
    return None


_STEM = {"v8": {"n": 16, "s": 32, "m": 48, "l": 64, "x": 80}, "11": {"n": 16, "s": 32, "m": 64, "l": 64, "x": 96}}


def guess_variant_family(state: Dict[str, torch.Tensor], family: str) -> Optional[str]:
    if "model.0.conv.weight" not in state:
        return None
    c0 = int(state["model.0.conv.weight"].shape[0])
    cands = [v for v, c in _STEM[family].items() if c == c0]
    if len(cands) > 1:                       # yolo11 m / l share the widths, l repeats every block twice
        n2 = len({k.split(".")[3] for k in state if k.startswith("model.2.m.")})
        cands = [v for v in cands if (2 if v in "lx" else 1) == n2] or cands
    return cands[0] if cands else None


def guess_variant(state: Dict[str, torch.Tensor]) -> Optional[str]:
    """Identify n/s/m/b/l/x from the stem width and the layer-6 / layer-2 repeat structure."""
    if "model.0.conv.weight" not in state:
        return None
    c0 = int(state["model.0.conv.weight"].shape[0])
    n2 = len({k.split(".")[3] for k in state if k.startswith("model.2.m.")})
    cands = []
    for v in "nsmblx":
        ch, reps = resolve_channels(v)
        if ch[0] == c0 and reps[2] == n2:
            cands.append(v)
    if len(cands) > 1:        # b vs l share widths and differ in depth only
        n4 = len({k.split(".")[3] for k in state if k.startswith("model.4.m.")})
        cands = [v for v in cands if resolve_channels(v)[1][4] == n4] or cands
    return cands[0] if cands else None


# --------------------------------------------------------------------------------------------------------
# BN folding -> what the engine stores:  name -> (weight [Cout,Cin/g,kh,kw] fp32, bias [Cout] fp32)
# --------------------------------------------------------------------------------------------------------
def fold_state(state: Dict[str, torch.Tensor]) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """A.1: W' = W*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps); RepVGGDW (A.2) merged into one
    7x7 depthwise; biased Conv2d/ConvTranspose2d passed through. Training-only one-to-many branches dropped."""
    out: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
    for k, w in state.items():
        if k.endswith(".conv.weight") and (k[: -len("conv.weight")] + "bn.weight") in state:
            p = k[: -len(".conv.weight")]
            g, b = state[f"{p}.bn.weight"].float(), state[f"{p}.bn.bias"].float()
            m, v = state[f"{p}.bn.running_mean"].float(), state[f"{p}.bn.running_var"].float()
            scale = g / torch.sqrt(v + BN_EPS)
            out[p] = (w.float() * scale.view(-1, 1, 1, 1), b - m * scale)
        elif k.endswith(".weight") and not k.endswith(".bn.weight") and (k[:-7] + ".bias") in state \
                and w.dim() == 4:
            out[k[:-7]] = (w.float(), state[k[:-7] + ".bias"].float())
    for n in list(out):
        if n.endswith(".conv1") and n[:-1] in out:
            w7, b7 = out.pop(n[:-1])
            w3, b3 = out.pop(n)
            out[n[: -len(".conv1")]] = (w7 + torch.nn.functional.pad(w3, [2, 2, 2, 2]), b7 + b3)
    if any(".one2one_cv2." in n for n in out):
        for n in list(out):
            if n.startswith("model.23.cv2.") or n.startswith("model.23.cv3."):
                del out[n]                  # v10: one-to-many twins never reach the result (README.md:25)
    return out


# --------------------------------------------------------------------------------------------------------
# synthetic deterministic weights (no checkpoint exists offline)
# --------------------------------------------------------------------------------------------------------
def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return g


def synthetic_state(variant: str = "s", nc: int = 80, seg: bool = False, seed: int = 0,
                    cls_bias: Optional[float] = None, gain: float = 1.0, head_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Unfused ultralytics-layout state dict with seeded values:
    conv ~ N(0, sqrt(2/fan_in)); BN gamma=1, beta~U(-0.1,0.1), mean=0, var=1 (eps 1e-3 applied at fold);
    head biases as ultralytics `bias_init`: box 1.0, cls log(5/nc/(640/stride)^2) (or `cls_bias` if given)."""
    st: Dict[str, torch.Tensor] = {}

    def conv_bn(p: str, c1: int, c2: int, k: int, g: int = 1):
        fan_in = (c1 // g) * k * k
        st[f"{p}.conv.weight"] = torch.randn((c2, c1 // g, k, k), generator=_gen(p + ".w", seed)) * (gain * math.sqrt(2.0 / fan_in))
        st[f"{p}.bn.weight"] = torch.ones(c2)
        st[f"{p}.bn.bias"] = (torch.rand(c2, generator=_gen(p + ".b", seed)) - 0.5) * 0.2
        st[f"{p}.bn.running_mean"] = torch.zeros(c2)
        st[f"{p}.bn.running_var"] = torch.ones(c2)

    def conv_bias(p: str, c1: int, c2: int, k: int, bias: torch.Tensor, transpose: bool = False):
        fan_in = c1 * k * k
        shape = (c1, c2, k, k) if transpose else (c2, c1, k, k)
        st[f"{p}.weight"] = torch.randn(shape, generator=_gen(p + ".w", seed)) * (head_gain * math.sqrt(1.0 / fan_in))
        st[f"{p}.bias"] = bias

    for op in layer_plan(variant):
        kind, p = op["kind"], op["name"]
        if kind == "conv":
            conv_bn(p, op["c1"], op["c2"], op["k"])
        elif kind in ("c2f", "c2fcib"):
            c, n = op["c"], op["n"]
            conv_bn(f"{p}.cv1", op["c1"], 2 * c, 1)
            conv_bn(f"{p}.cv2", (2 + n) * c, op["c2"], 1)
            for j in range(n):
                q = f"{p}.m.{j}"
                if kind == "c2f":
                    conv_bn(f"{q}.cv1", c, c, 3)
                    conv_bn(f"{q}.cv2", c, c, 3)
                else:
                    conv_bn(f"{q}.cv1.0", c, c, 3, g=c)
                    conv_bn(f"{q}.cv1.1", c, 2 * c, 1)
                    if op["lk"]:
                        conv_bn(f"{q}.cv1.2.conv", 2 * c, 2 * c, 7, g=2 * c)
                        conv_bn(f"{q}.cv1.2.conv1", 2 * c, 2 * c, 3, g=2 * c)
                    else:
                        conv_bn(f"{q}.cv1.2", 2 * c, 2 * c, 3, g=2 * c)
                    conv_bn(f"{q}.cv1.3", 2 * c, c, 1)
                    conv_bn(f"{q}.cv1.4", c, c, 3, g=c)
        elif kind == "scdown":
            conv_bn(f"{p}.cv1", op["c1"], op["c2"], 1)
            conv_bn(f"{p}.cv2", op["c2"], op["c2"], op["k"], g=op["c2"])
        elif kind == "sppf":
            conv_bn(f"{p}.cv1", op["c1"], op["c1"] // 2, 1)
            conv_bn(f"{p}.cv2", 2 * op["c1"], op["c2"], 1)
        elif kind == "psa":
            c, nh, kd = op["c"], op["nh"], op["kd"]
            conv_bn(f"{p}.cv1", op["c1"], 2 * c, 1)
            conv_bn(f"{p}.cv2", 2 * c, op["c1"], 1)
            conv_bn(f"{p}.attn.qkv", c, c + 2 * kd * nh, 1)
            conv_bn(f"{p}.attn.proj", c, c, 1)
            conv_bn(f"{p}.attn.pe", c, c, 3, g=c)
            conv_bn(f"{p}.ffn.0", c, 2 * c, 1)
            conv_bn(f"{p}.ffn.1", 2 * c, c, 1)
    chs, c2, c3, c4 = head_dims(variant, nc)
    p = "model.23"
    for l, (x, s) in enumerate(zip(chs, (8, 16, 32))):
        conv_bn(f"{p}.one2one_cv2.{l}.0", x, c2, 3)
        conv_bn(f"{p}.one2one_cv2.{l}.1", c2, c2, 3)
        conv_bias(f"{p}.one2one_cv2.{l}.2", c2, 4 * REG_MAX, 1, torch.full((4 * REG_MAX,), 1.0))
        conv_bn(f"{p}.one2one_cv3.{l}.0.0", x, x, 3, g=x)
        conv_bn(f"{p}.one2one_cv3.{l}.0.1", x, c3, 1)
        conv_bn(f"{p}.one2one_cv3.{l}.1.0", c3, c3, 3, g=c3)
        conv_bn(f"{p}.one2one_cv3.{l}.1.1", c3, c3, 1)
        cb = math.log(5 / nc / (640 / s) ** 2) if cls_bias is None else cls_bias
        conv_bias(f"{p}.one2one_cv3.{l}.2", c3, nc, 1, torch.full((nc,), cb))
        if seg:
            conv_bn(f"{p}.cv4.{l}.0", x, c4, 3)
            conv_bn(f"{p}.cv4.{l}.1", c4, c4, 3)
            conv_bias(f"{p}.cv4.{l}.2", c4, NM, 1, torch.zeros(NM))
    if seg:
        npr = chs[0]
        conv_bn(f"{p}.proto.cv1", chs[0], npr, 3)
        conv_bias(f"{p}.proto.upsample", npr, npr, 2,
                  (torch.rand(npr, generator=_gen(p + ".proto.up.b", seed)) - 0.5) * 0.2, transpose=True)
        conv_bn(f"{p}.proto.cv2", npr, npr, 3)
        conv_bn(f"{p}.proto.cv3", npr, NM, 1)
    return st


def synthetic_state_family(family: str, variant: str = "n", nc: int = 80, seed: int = 0, cls_bias: Optional[float] = None,
                           gain: float = 1.0, head_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded unfused state dict of a YOLOv8-seg / YOLO11-seg checkpoint. The names and shapes come from the engine's own graph
    (yp_weight_info: no second copy of the layer arithmetic on the host); plain biased convs (the last conv of every head branch,
    Proto's ConvTranspose) stay plain, everything else becomes conv + BatchNorm as ultralytics stores it."""
    from .engine import Engine
    hi = {"v8": 22, "11": 23}[family]
    eng = Engine(variant, nc, True, "bf16", 0, family=family)
    table = eng.expected_weights()
    eng.close()
    st: Dict[str, torch.Tensor] = {}
    for name, shape in table:
        if not name.endswith(".weight"):
            continue
        p = name[:-7]
        plain = p.endswith(".2") and p.startswith(f"model.{hi}.cv") or p.endswith(".proto.upsample")
        if plain:
            transpose = p.endswith(".upsample")
            cin = shape[0] if transpose else shape[1]
            fan_in = cin * shape[2] * shape[3]
            st[f"{p}.weight"] = torch.randn(shape, generator=_gen(p + ".w", seed)) * (head_gain * math.sqrt(1.0 / fan_in))
            cout = shape[1] if transpose else shape[0]
            if f".cv2." in p:
                b = torch.full((cout,), 1.0)
            elif f".cv3." in p:
                l = int(p.split(".")[3])
                b = torch.full((cout,), math.log(5 / nc / (640 / (8 << l)) ** 2) if cls_bias is None else cls_bias)
            elif p.endswith(".upsample"):
                b = (torch.rand(cout, generator=_gen(p + ".b", seed)) - 0.5) * 0.2
            else:
                b = torch.zeros(cout)
            st[f"{p}.bias"] = b
        else:
            c2, c1g, k, _ = shape
            fan_in = c1g * k * k
            st[f"{p}.conv.weight"] = torch.randn(shape, generator=_gen(p + ".w", seed)) * (gain * math.sqrt(2.0 / fan_in))
            st[f"{p}.bn.weight"] = torch.ones(c2)
            st[f"{p}.bn.bias"] = (torch.rand(c2, generator=_gen(p + ".b", seed)) - 0.5) * 0.2
            st[f"{p}.bn.running_mean"] = torch.zeros(c2)
            st[f"{p}.bn.running_var"] = torch.ones(c2)
    st[f"model.{hi}.dfl.conv.weight"] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
    return st


def save_as_ultralytics_pt(state: Dict[str, torch.Tensor], path: str) -> None:
    """Test helper: write a checkpoint in the *dict-of-tensors* form `{"model": state_dict}` (fp16 as released
    weights are). The full-module-pickle form is exercised in tests with fabricated stub modules."""
    torch.save({"model": {k: v.half() for k, v in state.items()}, "ema": None, "version": "synthetic"}, path)
