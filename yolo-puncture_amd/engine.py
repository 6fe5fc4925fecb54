"""ctypes binding of libyolop.so (include/yolop.h). Tensors in / tensors out; PyTorch is only the owner of device
memory and streams. There is NO fallback: if the HIP library is missing or no GPU is present, calls raise.

Replaces what `ultralytics.YOLO(path)` builds and what `.predict` runs (reference yolo_seg/app.py:45,91).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .weights import fold_state

_LIB: Optional[C.CDLL] = None
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YOLOP_LIB") or os.path.join(_HERE, "libyolop.so")     # YOLOP_LIB: another build of the same ABI (A/B runs)

YP_BF16, YP_F32 = 0, 1
TASK_DETECT, TASK_SEGMENT = 0, 1
OP_KINDS = {0: "stem", 1: "conv", 2: "dwconv", 3: "pool5", 4: "upsample", 5: "attn", 6: "head", 7: "convT", 8: "pool3", 9: "amax"}


class ModelDesc(C.Structure):
    _fields_ = [("variant", C.c_int), ("nc", C.c_int), ("task", C.c_int), ("dtype", C.c_int), ("max_det", C.c_int), ("family", C.c_int)]


class YolopError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libyolop.so and declare the prototypes of include/yolop.h. Raises if it was not built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.isfile(p):
        raise YolopError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
    lib = C.CDLL(p)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    lib.yp_last_error.restype = C.c_char_p
    lib.yp_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.POINTER(vp)]
    lib.yp_destroy.argtypes = [vp]
    lib.yp_weight_count.argtypes = [vp]
    lib.yp_weight_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), ip]
    lib.yp_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), C.c_int]
    lib.yp_finalize.argtypes = [vp]
    lib.yp_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.yp_proto.argtypes = [vp, C.POINTER(vp), ip, ip]
    lib.yp_masks.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.yp_id_mask_resized.argtypes = [vp, C.c_int, vp, vp] + [C.c_int] * 5 + [vp, vp, C.c_int, C.c_int, vp]
    lib.yp_id_mask_resized.restype = C.c_int
    lib.yp_mask_contours.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.yp_mask_contours.restype = C.c_int
    lib.yp_plan.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.yp_op_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, ip, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.yp_op_output.argtypes = [vp, C.c_int, ip, ip, ip]
    lib.yp_op_kernel.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
    lib.yp_op_fusion.argtypes = [vp, C.c_int, ip, ip]
    lib.yp_op_fusion.restype = C.c_int
    lib.yp_run_op.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.yp_tensor_write.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.yp_tensor_count.argtypes = [vp]
    lib.yp_tensor_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, ip, ip]
    lib.yp_tensor_read.argtypes = [vp, C.c_int, vp]
    lib.yp_profile.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.yp_set_graph.argtypes = [vp, C.c_int]
    lib.yp_set_autotune.argtypes = [vp, C.c_int]
    lib.yp_set_nms.argtypes = [vp, C.c_float, C.c_float]
    lib.yp_set_nms.restype = C.c_int
    lib.yp_debug_host_selftest.argtypes = [vp]
    lib.yp_debug_host_selftest.restype = C.c_int
    lib.yp_tuning_source.argtypes = [vp]
    lib.yp_tuning_source.restype = C.c_int
    lib.yp_debug_graph_info.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.yp_debug_graph_info.restype = C.c_int
    lib.yp_debug_head_positions.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.yp_debug_head_positions.restype = C.c_int
    lib.yp_debug_marker.argtypes = [vp]
    lib.yp_debug_marker.restype = C.c_int
    lib.yp_debug_force_conv_cfg.argtypes = [C.c_int]
    lib.yp_debug_ablation.argtypes = [C.c_int]
    lib.yp_debug_head_clocks.argtypes = [C.POINTER(C.c_uint64)]
    lib.yp_debug_head_branch_clocks.argtypes = [C.POINTER(C.c_uint64)]
    lib.yp_debug_contour_clocks.argtypes = [C.POINTER(C.c_uint64)]
    lib.yp_debug_pwsp_clocks.argtypes = [C.POINTER(C.c_uint64)]
    lib.yp_letterbox.argtypes = [vp, C.c_int, C.c_int, vp] + [C.c_int] * 7 + [vp]
    lib.yp_letterbox.restype = C.c_int
    lib.yp_comm_unique_id.argtypes = [vp]
    lib.yp_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.yp_allgather.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.yp_comm_destroy.argtypes = [vp]
    for fn in ("yp_comm_unique_id", "yp_comm_create", "yp_allgather", "yp_comm_destroy"):
        getattr(lib, fn).restype = C.c_int
    for fn in ("yp_create", "yp_destroy", "yp_weight_count", "yp_weight_info", "yp_set_weight", "yp_finalize",
               "yp_forward", "yp_proto", "yp_masks", "yp_plan", "yp_op_info", "yp_op_output", "yp_op_input", "yp_tensor_count",
               "yp_tensor_info", "yp_tensor_read", "yp_profile", "yp_set_graph", "yp_run_op", "yp_tensor_write",
               "yp_op_kernel", "yp_set_autotune", "yp_debug_force_conv_cfg", "yp_debug_ablation"):
        getattr(lib, fn).restype = C.c_int
    if path is None:
        _LIB = lib
    return lib


EXPORTS = ["yp_last_error", "yp_create", "yp_destroy", "yp_weight_count", "yp_weight_info", "yp_set_weight",
           "yp_finalize", "yp_forward", "yp_proto", "yp_masks", "yp_id_mask_resized", "yp_plan", "yp_op_info", "yp_op_output", "yp_op_input", "yp_op_fusion",
           "yp_tensor_count", "yp_tensor_info", "yp_tensor_read", "yp_profile", "yp_set_graph", "yp_run_op",
           "yp_tensor_write", "yp_op_kernel", "yp_set_autotune", "yp_tuning_export", "yp_tuning_import", "yp_set_nms", "yp_debug_force_conv_cfg", "yp_debug_ablation", "yp_debug_head_clocks", "yp_debug_head_branch_clocks", "yp_debug_head_winners", "yp_debug_contour_clocks", "yp_debug_pwsp_clocks", "yp_debug_host_selftest", "yp_debug_graph_info", "yp_tuning_source", "yp_debug_head_positions", "yp_debug_marker", "yp_letterbox", "yp_mask_contours",
           "yp_comm_unique_id", "yp_comm_create", "yp_allgather", "yp_comm_destroy",
           "yp_u2net_create", "yp_u2net_destroy", "yp_u2net_weight_count", "yp_u2net_weight_info", "yp_u2net_set_weight", "yp_u2net_finalize",
           "yp_u2net_forward", "yp_u2net_set_graph", "yp_u2net_tensor_count", "yp_u2net_tensor_info", "yp_u2net_tensor_read"]


def _stream_ptr(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def letterbox_device(src: torch.Tensor, geo: dict, out: Optional[torch.Tensor] = None, pad_value: int = 114) -> torch.Tensor:
    """LetterBox on the GPU (yp_letterbox): `src` uint8 cuda [h0,w0,3], `geo` from hostops.letterbox_geometry ->
    uint8 cuda [out_h,out_w,3] (written into `out` when given, e.g. a row of the batch tensor)."""
    if not (src.is_cuda and src.dtype == torch.uint8 and src.dim() == 3 and src.shape[2] == 3 and src.is_contiguous()):
        raise ValueError("letterbox_device needs a contiguous uint8 CUDA tensor [H,W,3]")
    oh, ow = int(geo["out_h"]), int(geo["out_w"])
    if out is None:
        out = torch.empty((oh, ow, 3), dtype=torch.uint8, device=src.device)
    if not (out.is_cuda and out.dtype == torch.uint8 and tuple(out.shape) == (oh, ow, 3) and out.is_contiguous()):
        raise ValueError(f"letterbox_device: output must be a contiguous uint8 CUDA tensor [{oh},{ow},3]")
    lib = load_library()
    with torch.cuda.device(src.device):
        rc = lib.yp_letterbox(C.c_void_p(src.data_ptr()), int(src.shape[0]), int(src.shape[1]), C.c_void_p(out.data_ptr()), oh, ow,
                              int(geo["new_h"]), int(geo["new_w"]), int(geo["top"]), int(geo["left"]), int(pad_value),
                              C.c_void_p(_stream_ptr(src.device)))
    if rc != 0:
        raise YolopError(lib.yp_last_error().decode())
    return out


CONTOUR_STRATEGIES = {"largest": 0, "all": 1}      # include/yolop.h YP_CONTOURS_*
_PARTS_CAP = 65                                     # [count | up to 64 contour lengths] (csrc/contour.hip CT_NLMAX)


def mask_contours_device(masks: torch.Tensor, max_pts: Optional[int] = None, want_rect: bool = True, strategy: str = "all", want_parts: bool = False):
    """yp_mask_contours: uint8 cuda [n,H,W] -> (list of int32 [m,2] numpy polygons (None where the device path declined), rect float64 [n,2]
    numpy (long side, short side) or None[, list of per-mask contour lengths when `want_parts`]). `strategy` as ultralytics' masks2segments:
    "all" (every external contour, concatenated bottom-up) or "largest". The masks stay on the device; counts, contour lengths, rectangles
    and the heads of the point lists share one allocation so that a single mask (what the reference's loop asks for per frame) costs ONE
    device-to-host copy. `max_pts` also sizes the kernel's per-candidate lists (1024 points each behind the result): the default is
    generous for one mask, 16384 per mask otherwise."""
    if not (masks.is_cuda and masks.dtype == torch.uint8 and masks.dim() == 3):
        raise ValueError("mask_contours_device needs a uint8 CUDA tensor [n,H,W]")
    if strategy not in CONTOUR_STRATEGIES:
        raise ValueError(f"strategy must be 'all' or 'largest', got {strategy!r}")
    masks = masks.contiguous()
    n, H, W = (int(v) for v in masks.shape)
    if max_pts is None:
        max_pts = 131072 if n <= 2 else 16384
    dev = masks.device
    # int32 words: [count (n) | pad | parts (n x _PARTS_CAP, padded to even) | rect (n x 2 float64) | points (n x max_pts x 2)]
    o_parts = (n + 1) // 2 * 2
    o_rect = o_parts + (n * _PARTS_CAP + 1) // 2 * 2
    o_pts = o_rect + 4 * n
    buf = torch.empty((o_pts + n * max_pts * 2,), dtype=torch.int32, device=dev)
    lib = load_library()
    with torch.cuda.device(dev):
        base = buf.data_ptr()
        rc = lib.yp_mask_contours(C.c_void_p(masks.data_ptr()), n, H, W, CONTOUR_STRATEGIES[strategy], int(max_pts), C.c_void_p(base + 4 * o_pts),
                                  C.c_void_p(base), C.c_void_p(base + 4 * o_parts), _PARTS_CAP, C.c_void_p(base + 4 * o_rect if want_rect else None),
                                  C.c_void_p(_stream_ptr(dev)))
    if rc != 0:
        raise YolopError(lib.yp_last_error().decode())
    if n == 0:
        return ([], (np.zeros((0, 2)) if want_rect else None), []) if want_parts else ([], (np.zeros((0, 2)) if want_rect else None))
    head_pts = min(max_pts, 1024)
    if n == 1:
        h = buf[:o_pts + 2 * head_pts].cpu().numpy()
        c = h[:1]
        if c[0] > head_pts:
            host = buf[o_pts:o_pts + 2 * int(c[0])].cpu().numpy().reshape(1, -1, 2)
        else:
            host = h[o_pts:].reshape(1, -1, 2)
    else:
        h = buf[:o_pts].cpu().numpy()
        c = h[:n]
        top = int(max(1, c.max()))
        host = buf[o_pts:].view(n, max_pts, 2)[:, :top].cpu().numpy()
    rect = h[o_rect:o_rect + 4 * n].view(np.float64).reshape(n, 2).copy() if want_rect else None
    polys = [host[i, :c[i]].copy() if c[i] >= 0 else None for i in range(n)]
    if not want_parts:
        return polys, rect
    pr = h[o_parts:o_parts + n * _PARTS_CAP].reshape(n, _PARTS_CAP)
    parts = [[int(v) for v in pr[i, 1:1 + min(int(pr[i, 0]), _PARTS_CAP - 1)]] if c[i] >= 0 else None for i in range(n)]
    return polys, rect, parts


class Engine:
    """One engine per GPU. `state` is an (unfused) ultralytics-layout state dict; see weights.py."""

    def __init__(self, variant: str = "s", nc: int = 80, seg: bool = False, dtype: str = "bf16",
                 device: int = 0, max_det: int = 300, state: Optional[Dict[str, torch.Tensor]] = None,
                 finalize: bool = True, family: str = "v10"):
        self.lib = load_library()
        self.variant, self.nc, self.seg, self.max_det = variant, nc, seg, max_det
        self.family = family
        if family != "v10" and not seg:
            raise ValueError("the v8 / 11 families are built as segmentation models (the checkpoints the reference ships)")
        self.dtype = {"bf16": YP_BF16, "fp32": YP_F32, "f32": YP_F32}[dtype]
        self.device_index = int(device)
        self._h = C.c_void_p()
        desc = ModelDesc(ord(variant), nc, TASK_SEGMENT if seg else TASK_DETECT, self.dtype, max_det, {"v10": 0, "v8": 8, "11": 11}[family])
        self._chk(self.lib.yp_create(C.byref(desc), self.device_index, C.byref(self._h)))
        self._keep: List[torch.Tensor] = []
        self.finalized = False
        if state is not None:
            self.load_state(state)
            if finalize:
                self.finalize()

    # -- errors ------------------------------------------------------------------------------------------------
    def _chk(self, rc: int) -> int:
        if rc < 0:
            raise YolopError(f"libyolop error {rc}: {self.lib.yp_last_error().decode()}")
        return rc

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.yp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -----------------------------------------------------------------------------------------------
    def expected_weights(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out = []
        n = self._chk(self.lib.yp_weight_count(self._h))
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(n):
            self._chk(self.lib.yp_weight_info(self._h, i, name, 256, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        return out

    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        """Fold Conv+BN (and merge RepVGGDW) on the host, hand every folded fp32 tensor to the engine."""
        folded = fold_state(state)
        for name, (w, b) in folded.items():
            for suffix, t in ((".weight", w), (".bias", b)):
                t = t.detach().to(torch.float32).contiguous().cpu()
                shp = (C.c_int64 * t.dim())(*t.shape)
                self._chk(self.lib.yp_set_weight(self._h, (name + suffix).encode(), C.c_void_p(t.data_ptr()), shp, t.dim()))

    def finalize(self) -> None:
        self._chk(self.lib.yp_finalize(self._h))
        self.finalized = True

    def set_autotune(self, enable: bool) -> None:
        self._chk(self.lib.yp_set_autotune(self._h, 1 if enable else 0))

    def tuning_export(self) -> List[int]:
        """Tile configuration id per op of the current plan (after a forward): what `tuning_import` takes on another engine / rank."""
        n = self._chk(self.lib.yp_tuning_export(self._h, None, 0))
        buf = (C.c_int32 * n)()
        self._chk(self.lib.yp_tuning_export(self._h, buf, n))
        return list(buf)

    def tuning_import(self, B: int, H: int, W: int, cfgs) -> None:
        """Install another engine's tile configurations for shape (B,H,W): the next forward of that shape does not tune, and its bf16
        results equal the exporting engine's bit for bit (same build)."""
        cfgs = [int(c) for c in cfgs]
        buf = (C.c_int32 * len(cfgs))(*cfgs)
        self._chk(self.lib.yp_tuning_import(self._h, int(B), int(H), int(W), buf, len(cfgs)))

    def set_nms(self, conf: float = 0.25, iou: float = 0.7) -> None:
        """families v8 / 11: thresholds of the NMS inside the forward (`.predict(conf=, iou=)`)."""
        self._chk(self.lib.yp_set_nms(self._h, float(conf), float(iou)))

    def set_graph(self, enable) -> None:
        """False / 0: eager launches; True / 1: hipGraph replay with head lanes; 2: replay without lanes; "auto" / 3: per input shape
        whichever of eager and replay a one-off timing finds faster (include/yolop.h yp_set_graph)."""
        mode = 3 if enable == "auto" else int(enable)
        self._chk(self.lib.yp_set_graph(self._h, mode))

    # -- hot path ------------------------------------------------------------------------------------------------
    def forward(self, im: torch.Tensor, out: Optional[dict] = None) -> dict:
        """im: uint8 cuda tensor [B,H,W,3] BGR, letterboxed (H,W multiples of 32).
        -> dict(det [B,max_det,6] f32, idx [B,max_det] i32, coeff [B,max_det,32] f32 | None)."""
        if not im.is_cuda or im.dtype != torch.uint8 or im.dim() != 4 or im.shape[-1] != 3:
            raise TypeError("forward expects a uint8 CUDA tensor [B,H,W,3]")
        if im.device.index != self.device_index:
            raise ValueError(f"input is on cuda:{im.device.index}, engine on cuda:{self.device_index}")
        im = im.contiguous()
        B, H, W, _ = im.shape
        dev = im.device
        if out is None:
            out = dict(det=torch.empty((B, self.max_det, 6), dtype=torch.float32, device=dev),
                       idx=torch.empty((B, self.max_det), dtype=torch.int32, device=dev),
                       coeff=torch.empty((B, self.max_det, 32), dtype=torch.float32, device=dev) if self.seg else None)
        cf = out["coeff"].data_ptr() if out.get("coeff") is not None else None
        self._chk(self.lib.yp_forward(self._h, C.c_void_p(im.data_ptr()), B, H, W, C.c_void_p(out["det"].data_ptr()),
                                      C.c_void_p(out["idx"].data_ptr()), C.c_void_p(cf), C.c_void_p(_stream_ptr(dev))))
        self._last_im = im    # keep the input alive until the stream has consumed it
        return out

    def proto(self) -> torch.Tensor:
        """Engine-owned prototypes of the last forward as fp32 [B,Hp,Wp,32] (a copy; test/debug helper)."""
        for t in self.tensors():
            if t["name"].endswith(".proto.cv3"):
                return self.read_tensor(t["index"])
        raise KeyError("proto.cv3")

    def masks(self, b: int, coeff: torch.Tensor, boxes: torch.Tensor, out_hw: Tuple[int, int], retina: bool = True,
              want_masks: bool = True, want_ids: bool = False, suppress_small: bool = False, min_area: int = 100):
        """-> (masks uint8 [n,oh,ow] | None, ids int64 [oh,ow] | None, kept int32 [n] | None)"""
        n = int(coeff.shape[0])
        dev = coeff.device
        oh, ow = int(out_hw[0]), int(out_hw[1])
        coeff = coeff.to(torch.float32).contiguous()
        boxes = boxes.to(torch.float32).contiguous()
        m = torch.empty((n, oh, ow), dtype=torch.uint8, device=dev) if want_masks else None
        ids = torch.empty((oh, ow), dtype=torch.int64, device=dev) if want_ids else None
        kept = torch.empty((n,), dtype=torch.int32, device=dev) if want_ids else None
        self._chk(self.lib.yp_masks(self._h, b, C.c_void_p(coeff.data_ptr()), C.c_void_p(boxes.data_ptr()), n, oh, ow,
                                    1 if retina else 0, C.c_void_p(m.data_ptr() if m is not None else None),
                                    C.c_void_p(ids.data_ptr() if ids is not None else None),
                                    C.c_void_p(kept.data_ptr() if kept is not None else None),
                                    1 if suppress_small else 0, int(min_area), C.c_void_p(_stream_ptr(dev))))
        self._keep = [coeff, boxes]
        return m, ids, kept

    def id_mask_resized(self, b: int, coeff: torch.Tensor, boxes: torch.Tensor, mask_hw: Tuple[int, int], out_hw: Tuple[int, int],
                        suppress_small: bool = False, min_area: int = 100):
        """`auto_segment` with min_side > 0 (yp_id_mask_resized): masks at `mask_hw` (the shrunk frame), antialiased bilinear to
        `out_hw`, float-area test, paint. -> (ids int64 [out_hw] cuda, kept int32 [n] cuda)"""
        n = int(coeff.shape[0])
        dev = coeff.device
        coeff = coeff.to(torch.float32).contiguous()
        boxes = boxes.to(torch.float32).contiguous()
        ids = torch.empty((int(out_hw[0]), int(out_hw[1])), dtype=torch.int64, device=dev)
        kept = torch.empty((n,), dtype=torch.int32, device=dev)
        self._chk(self.lib.yp_id_mask_resized(self._h, b, C.c_void_p(coeff.data_ptr()), C.c_void_p(boxes.data_ptr()), n, int(mask_hw[0]),
                                              int(mask_hw[1]), int(out_hw[0]), int(out_hw[1]), C.c_void_p(ids.data_ptr()),
                                              C.c_void_p(kept.data_ptr()), 1 if suppress_small else 0, int(min_area),
                                              C.c_void_p(_stream_ptr(dev))))
        self._keep = [coeff, boxes]
        return ids, kept

    # -- introspection ---------------------------------------------------------------------------------------------
    def plan(self, B: int, H: int, W: int) -> List[dict]:
        n = self._chk(self.lib.yp_plan(self._h, B, H, W))
        ops = []
        name = C.create_string_buffer(256)
        kind, fl, by = C.c_int(), C.c_double(), C.c_double()
        t, co, cc = C.c_int(), C.c_int(), C.c_int()
        for i in range(n):
            self._chk(self.lib.yp_op_info(self._h, i, name, 256, C.byref(kind), C.byref(fl), C.byref(by)))
            self._chk(self.lib.yp_op_output(self._h, i, C.byref(t), C.byref(co), C.byref(cc)))
            rec = dict(name=name.value.decode(), kind=OP_KINDS[kind.value], flops=fl.value, bytes=by.value,
                       out=(t.value, co.value, cc.value))
            self._chk(self.lib.yp_op_kernel(self._h, i, name, 256))
            rec["kernel"] = name.value.decode()
            cr = C.c_int()
            self._chk(self.lib.yp_op_input(self._h, i, C.byref(t), C.byref(co), C.byref(cc), C.byref(cr)))
            rec["in"] = (t.value, co.value, cc.value)
            rec["c_read"] = cr.value
            pre, stored = C.c_int(), C.c_int()
            self._chk(self.lib.yp_op_fusion(self._h, i, C.byref(pre), C.byref(stored)))
            rec["pre"], rec["pre_stored"] = pre.value, bool(stored.value)     # pwsp_kernel: the 1x1 conv fused in front (or -1), and whether its output is written too
            ops.append(rec)
        return ops

    def tensors(self) -> List[dict]:
        n = self._chk(self.lib.yp_tensor_count(self._h))
        name = C.create_string_buffer(256)
        dims = (C.c_int * 4)()
        f32 = C.c_int()
        out = []
        for i in range(n):
            self._chk(self.lib.yp_tensor_info(self._h, i, name, 256, dims, C.byref(f32)))
            out.append(dict(index=i, name=name.value.decode(), shape=tuple(dims), f32=bool(f32.value)))
        return out

    def find_tensor(self, name: str) -> int:
        for t in self.tensors():
            if t["name"] == name:
                return t["index"]
        raise KeyError(name)

    def read_tensor(self, index: int) -> torch.Tensor:
        """Debug tap: NHWC fp32 host copy of an engine-owned activation of the last forward."""
        info = self.tensors()[index]
        out = torch.empty(info["shape"], dtype=torch.float32)
        self._chk(self.lib.yp_tensor_read(self._h, index, C.c_void_p(out.data_ptr())))
        return out

    def run_op(self, i: int, im: torch.Tensor, out: dict) -> None:
        """Debug stepping: launch op i of the current plan (inputs are whatever the engine tensors hold)."""
        cf = out["coeff"].data_ptr() if out.get("coeff") is not None else None
        self._chk(self.lib.yp_run_op(self._h, i, C.c_void_p(im.data_ptr()), C.c_void_p(out["det"].data_ptr()),
                                     C.c_void_p(out["idx"].data_ptr()), C.c_void_p(cf), C.c_void_p(_stream_ptr(im.device))))

    def write_tensor(self, index: int, coff: int, data_nhwc: torch.Tensor) -> None:
        """Debug: overwrite channels [coff, coff+C) of an engine tensor from an fp32 host tensor [B,H,W,C]."""
        d = data_nhwc.detach().to(torch.float32).contiguous().cpu()
        self._chk(self.lib.yp_tensor_write(self._h, index, coff, int(d.shape[-1]), C.c_void_p(d.data_ptr())))

    def head_winners(self, B: int):
        """Debug: what the winners-only head left behind the last forward - (mode bits, sel [B,512] int32, box rows [B,max_det,64],
        coefficient rows [B,max_det,32]); mode 0 = dense head (arrays are then zeros)."""
        sel = torch.zeros((B, 512), dtype=torch.int32)
        box = torch.zeros((B, self.max_det, 64), dtype=torch.float32)
        cf = torch.zeros((B, self.max_det, 32), dtype=torch.float32)
        mode = self._chk(self.lib.yp_debug_head_winners(self._h, C.c_void_p(sel.data_ptr()), C.c_void_p(box.data_ptr()), C.c_void_p(cf.data_ptr())))
        return mode, sel, box, cf

    def tuning_source(self) -> str:
        """Where the current plan's tile configurations came from: "tuner", "cache file" (YOLOP_TUNE_CACHE) or "packaged table"."""
        r = self.lib.yp_tuning_source(self._h)
        if r < 0:
            self._chk(r)
        return ("tuner", "cache file", "packaged table")[r]

    def graph_info(self) -> dict:
        """What the last hipGraph capture built (yp_debug_graph_info): captures so far, nodes, edges, the lane schedule's edge count."""
        v = (C.c_int64 * 6)()
        self._chk(self.lib.yp_debug_graph_info(self._h, v))
        return dict(captures=int(v[0]), nodes=int(v[1]), edges=int(v[2]), schedule_edges=int(v[3]), lanes=int(v[4]), live=bool(v[5]))

    def head_positions(self) -> Optional[dict]:
        """Winners-only head, last forward: distinct 3x3-neighbourhood positions and winners per level (P3, P4, P5); None = dense head."""
        v = (C.c_int64 * 6)()
        if self._chk(self.lib.yp_debug_head_positions(self._h, v)) == 0:
            return None
        return dict(positions=[int(v[i]) for i in range(3)], winners=[int(v[3 + i]) for i in range(3)])

    def profile(self, im: torch.Tensor, iters: int = 5) -> List[dict]:
        """Per-op HIP-event timing (eager, one event pair per launch) on the current stream."""
        B, H, W, _ = im.shape
        ops = self.plan(B, H, W)
        dev = im.device
        det = torch.empty((B, self.max_det, 6), dtype=torch.float32, device=dev)
        idx = torch.empty((B, self.max_det), dtype=torch.int32, device=dev)
        cf = torch.empty((B, self.max_det, 32), dtype=torch.float32, device=dev) if self.seg else None
        ms = (C.c_float * len(ops))()
        self._chk(self.lib.yp_profile(self._h, C.c_void_p(im.data_ptr()), B, H, W, C.c_void_p(det.data_ptr()),
                                      C.c_void_p(idx.data_ptr()), C.c_void_p(cf.data_ptr() if cf is not None else None),
                                      ms, iters, C.c_void_p(_stream_ptr(dev))))
        for o, m in zip(ops, ms):
            o["ms"] = float(m)
        return ops
