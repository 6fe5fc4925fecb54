#!/usr/bin/env python3
"""Regenerates the committed golden vectors under tests/golden/ from the CPU oracle (oracle/ is a restatement of the
published YOLOv10 algorithm - the reference holds no fixtures for this path). Run from the repo root:
    python tests/golden/make_golden.py
Inputs are seeded; the .pt files hold inputs AND expected outputs so a test needs nothing else."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import make_case  # noqa: E402
from oracle.yolov10_oracle import Oracle, v10_postprocess  # noqa: E402
from oracle import postprocess_oracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # 1. end-to-end v10-N detect+seg on 2 frames of 64x96 (fp32 and bf16emu): detections, indices, coefficients, proto
    st, im = make_case("n", 80, True, 0, (2, 64, 96))
    case = dict(image=im, variant="n", nc=80, seg=True, seed=0, shape=(2, 64, 96))
    for mode in ("fp32", "bf16emu"):
        o = Oracle(st, "n", 80, True, mode).forward(im)
        case[mode] = dict(det=o["det"], idx=o["idx"].to(torch.int32), coeff=o["coeff"], proto=o["proto"].half())
    torch.save(case, os.path.join(OUT, "v10n_seg_64x96.pt"))

    # 2. top-k adversarial: exact ties, saturated scores, duplicates of one anchor, fewer anchors than max_det
    g = torch.Generator().manual_seed(1)
    scores = torch.rand(2, 40, 5, generator=g)
    scores[0, 3] = scores[0, 17]                      # two identical anchors -> lower index first
    scores[0, 7, :] = 1.0                             # saturated: all five classes of anchor 7 tie at 1.0
    scores[1, :, 2] = 0.5                             # a whole class column tied
    boxes = torch.rand(2, 40, 4, generator=g) * 100
    det, idx = v10_postprocess(boxes, scores, max_det=30)
    torch.save(dict(boxes=boxes, scores=scores, det=det, idx=idx.to(torch.int32), max_det=30), os.path.join(OUT, "topk_ties.pt"))

    # 3. mask tail: n=0,1,many ; box touching borders ; letterbox with padding (retina) and process_mask
    proto = torch.randn(32, 24, 40, generator=g)
    coeff = torch.randn(5, 32, generator=g)
    boxes_o = torch.tensor([[0., 0., 50., 30.], [10.5, 3.2, 117.9, 60.], [100., 20., 160., 90.], [30., 30., 31., 31.], [0., 0., 160., 90.]])
    m_native = po.process_mask_native(proto, coeff, boxes_o, (90, 160))
    boxes_in = boxes_o.clone()
    m_plain = po.process_mask(proto, coeff, boxes_in, (96, 160))
    ids, info = po.auto_segment_oracle(m_native, torch.tensor([.99, .97, .95, .93, .91]), torch.tensor([0., 1., 2., 3., 4.]),
                                       (90, 160), suppress_small_mask=True, min_area=100)
    torch.save(dict(proto=proto, coeff=coeff, boxes_orig=boxes_o, orig_hw=(90, 160), in_hw=(96, 160),
                    native=m_native.to(torch.uint8), plain=m_plain.to(torch.uint8), ids=ids, info=info),
               os.path.join(OUT, "mask_tail.pt"))
    print("golden written to", OUT)


if __name__ == "__main__":
    main()
