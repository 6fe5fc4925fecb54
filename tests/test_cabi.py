"""libyolop.so without a GPU: it loads, exports every symbol include/yolop.h declares, builds the same graph the
oracle restates (weights by name and shape, FLOPs per plan), and fails loudly in the states that cannot run."""
import ctypes as C
import os
import re

import pytest
import torch

from oracle.yolov10_oracle import count_conv_flops
from yolo_puncture_amd.engine import EXPORTS, Engine, ModelDesc, YolopError, load_library
from yolo_puncture_amd.weights import fold_state, synthetic_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "yolop.h")).read()
    declared = set(re.findall(r"\b(yp_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("yp_engine")
    lib = load_library()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/yolop.h but not exported"
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)


@pytest.mark.parametrize("v,seg", [("n", True), ("s", False), ("m", True), ("b", False), ("l", False), ("x", False)])
def test_graph_matches_folded_weights_and_flops(v, seg, monkeypatch):
    # the graph as the reference defines it: every branch of the head dense (the default plan evaluates the box / coefficient branches on the
    # top-k winners only and therefore executes fewer FLOPs: checked at the end)
    monkeypatch.setenv("YOLOP_DENSE_HEAD", "1")
    e = Engine(v, 80, seg, "bf16", 0)
    exp = dict(e.expected_weights())
    got = {}
    for k, (w, b) in fold_state(synthetic_state(v, 80, seg)).items():
        got[k + ".weight"], got[k + ".bias"] = tuple(w.shape), tuple(b.shape)
    assert exp == got
    ops = e.plan(1, 640, 640)
    assert abs(sum(o["flops"] for o in ops) - count_conv_flops(v, seg=seg)) < 1e6
    assert ops[0]["kind"] == "stem" and ops[-1]["kind"] == "head"
    # rectangular inputs re-plan with recomputed shapes (1280x720 letterboxes to 384x640)
    ops2 = e.plan(2, 384, 640)
    per_px, per_px2 = sum(o["flops"] for o in ops) / (640 * 640), sum(o["flops"] for o in ops2) / (2 * 384 * 640)
    assert abs(per_px2 - per_px) < 5e-3 * per_px          # (only the attention core, ~N^2, is not linear in the pixel count)
    # the host-only self-check of the executor (parameter blocks, kernel symbols, lane schedule invariants) for both plans
    assert e.lib.yp_debug_host_selftest(e._h) > 0
    e.plan(1, 640, 640)
    assert e.lib.yp_debug_host_selftest(e._h) > 0
    e.close()
    # default plan: winners-only head - the nine box-branch convolutions (and the coefficient branch's where its width is supported) launch
    # nothing, the head op carries their work on max_det winners per frame
    monkeypatch.delenv("YOLOP_DENSE_HEAD")
    e = Engine(v, 80, seg, "bf16", 0)
    ops3 = e.plan(1, 640, 640)
    box = [o for o in ops3 if ".one2one_cv2." in o["name"]]
    assert len(box) == 9
    if v == "x":      # 80-channel branch (padded taps): stays dense
        assert all(o["kernel"] != "-" for o in box) and "head_pos_kernel" not in ops3[-1]["kernel"]
    else:
        assert all(o["kernel"] == "-" and o["flops"] == 0 for o in box)
        assert "head_pos_kernel" in ops3[-1]["kernel"] and ops3[-1]["flops"] > 0
        assert sum(o["flops"] for o in ops3) < sum(o["flops"] for o in ops)
    assert e.lib.yp_debug_host_selftest(e._h) > 0
    e.close()


def test_errors_are_loud():
    lib = load_library()
    h = C.c_void_p()
    assert lib.yp_create(C.byref(ModelDesc(ord("q"), 80, 0, 0, 300)), 0, C.byref(h)) < 0
    assert b"variant" in lib.yp_last_error()
    e = Engine("n", 80, False, "bf16", 0)
    with pytest.raises(YolopError):
        e.plan(1, 100, 100)                           # not multiples of 32
    t = torch.zeros(3)
    shp = (C.c_int64 * 1)(3)
    assert lib.yp_set_weight(e._h, b"model.0.nonsense", C.c_void_p(t.data_ptr()), shp, 1) < 0
    assert lib.yp_set_weight(e._h, b"model.0.bias", C.c_void_p(t.data_ptr()), shp, 1) < 0      # wrong shape
    with pytest.raises(YolopError):
        e.finalize()                                  # weights missing
    e.load_state(synthetic_state("n", 80, False))
    if not torch.cuda.is_available():
        with pytest.raises(YolopError, match="no HIP device"):
            e.finalize()                              # no GPU here -> no silent CPU fallback
        assert lib.yp_forward(e._h, None, 1, 64, 64, None, None, None, None) < 0
    e.close()


def test_u2net_weight_table_equals_reference_module():
    """yp_u2net_create's own graph (C++ builder) names and shapes every parameter exactly as the REFERENCE nn.Module does
    (tests/golden/u2netp_params.npz was written from the reference's state_dict), after the conv+BatchNorm fold."""
    import numpy as np
    from yolo_puncture_amd.u2net import U2NetEngine, fold_state, synthetic_state
    z = np.load(os.path.join(ROOT, "tests", "golden", "u2netp_params.npz"))
    ref = {str(k): tuple(int(x) for x in str(s).split(",")) if str(s) else () for k, s in zip(z["names"], z["shapes"])}
    e = U2NetEngine("p", "fp32", 0)
    exp = dict(e.expected_weights())
    folded = {}
    for k, (w, b) in fold_state(synthetic_state("p", 0), "p").items():
        folded[k + ".weight"], folded[k + ".bias"] = tuple(w.shape), tuple(b.shape)
    assert exp == folded
    for k, shp in exp.items():                      # every engine parameter is a conv of the reference with that shape
        base, kind = k.rsplit(".", 1)
        rk = f"{base}.conv_s1.{kind}" if f"{base}.conv_s1.{kind}" in ref else k
        assert ref[rk] == shp, (k, rk)
    assert sum(1 for k in ref if k.endswith("conv_s1.weight") or k in ("outconv.weight",) or (k.startswith("side") and k.endswith(".weight"))) * 2 == len(exp)
    e.load_state(synthetic_state("p", 0))
    if not torch.cuda.is_available():
        with pytest.raises(YolopError, match="no HIP device"):
            e.finalize()
    e.close()
    with pytest.raises(YolopError, match="variant"):
        U2NetEngine("q", "fp32", 0)


def test_packaged_tune_tables_apply(monkeypatch):
    """Every table under yolo-puncture_amd/tune_tables names a (variant, task, batch, shape) whose plan accepts all of its tile-configuration
    ids in THIS build (the host self-check applies the table with the tuner's own validity predicates; no GPU needed). A configuration
    family that is renumbered or removed without bumping TUNE_TABLE_VERSION fails here, not on the GPU box."""
    import glob, os, re
    monkeypatch.delenv("YOLOP_TUNE_CACHE", raising=False)
    monkeypatch.delenv("YOLOP_NO_TUNE_TABLES", raising=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "yolo-puncture_amd", "tune_tables", "tt_*.txt")))
    assert files, "no packaged tune tables"
    for f in files:
        m = re.match(r"tt_f0(\w)(det|seg)_nc(\d+)_dt0_(\d+)x(\d+)x(\d+)_t\d+\.txt$", os.path.basename(f))
        assert m, f
        v, task, nc, B, H, W = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6))
        e = Engine(v, nc, task == "seg", "bf16", 0)
        ops = e.plan(B, H, W)
        names = {o["name"] for o in ops if o["kind"] in ("conv", "convT")}
        listed = {ln.split()[0] for ln in open(f) if ln.strip()}
        assert names <= listed, (f, sorted(names - listed)[:5])
        assert e.lib.yp_debug_host_selftest(e._h) > 0, (f, e.lib.yp_last_error())
        e.close()

