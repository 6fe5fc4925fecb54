"""`auto_segment` - the one function of the hot path that lives in the reference tree (yolo_seg/yolo_with_deva.py:37-88):
YOLO instance masks -> one int64 id mask + the per-object info list DEVA consumes. Same signature and behaviour; the
per-mask Python loop (`output_mask[mask > 0.5] = curr_id`, :62-86) runs as one GPU pass (yp_masks: GEMM, resize, crop,
threshold, area test, last-painter-wins id paint)."""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Tuple

import numpy as np
import torch



class ObjectInfo(NamedTuple):
    """Stand-in for deva.inference.object_info.ObjectInfo (external, not vendored): id, score, category_id."""
    id: int
    score: float
    category_id: int


def auto_segment(config: Dict, image: np.ndarray, yolo_model, min_side: int, suppress_small_mask: bool,
                 object_info_cls=ObjectInfo) -> Tuple[torch.Tensor, List]:
    """config: needs .get('MIN_AREA_THRESHOLD', 100); image: uint8 [h,w,3] (fed to YOLO as if BGR, exactly like the
    reference does with its RGB frames - SURVEY Appendix C-6); returns (int64 [h,w] on the model's device, [ObjectInfo])."""
    device = next(yolo_model.model.parameters()).device                     # :42
    h, w = image.shape[:2]
    pre = None
    if min_side > 0:                                                        # :45-48 (cv2.resize default = INTER_LINEAR), done on the GPU
        scale = min_side / min(h, w)
        pre = (int(w * scale), int(h * scale))
    min_area = config.get("MIN_AREA_THRESHOLD", 100) if hasattr(config, "get") else 100
    ids, kept, conf, cls = yolo_model.predict_id_mask(image, conf=0.9, out_hw=(h, w), suppress_small=suppress_small_mask,
                                                      min_area=int(min_area), pre_resize=pre)   # :45-79 fused
    segments_info = []
    kept = kept.tolist()
    for i, k in enumerate(kept):                                            # :82-86 ids consecutive over KEPT masks
        if k > 0:
            segments_info.append(object_info_cls(id=int(k), score=float(conf[i]), category_id=int(cls[i])))
    return ids.to(device), segments_info
