#!/usr/bin/env python3
"""Teacher-forced run up to one op; print where its output differs from the bf16-emulating oracle."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_case, nchw_to_nhwc
from oracle.yolov10_oracle import Oracle
from yolo_puncture_amd.engine import Engine
ap = argparse.ArgumentParser()
ap.add_argument("--op", required=True); ap.add_argument("--variant", default="s")
ap.add_argument("--shape", default="2,256,384"); ap.add_argument("--seg", action="store_true")
a = ap.parse_args()
shape = tuple(int(v) for v in a.shape.split(","))
st, im = make_case(a.variant, 80, a.seg, 0, shape)
taps = {}
Oracle(st, a.variant, 80, a.seg, "bf16emu", tap=lambda n, x: taps.__setitem__(n, x.float())).forward(im)
eng = Engine(a.variant, 80, a.seg, "bf16", 0, state=st)
imc = im.cuda()
out = eng.forward(imc)
torch.cuda.synchronize()
ops = eng.plan(*shape)
for i, o in enumerate(ops):
    if o["kind"] == "head":
        continue
    eng.run_op(i, imc, out)
    if o["name"] not in taps:
        continue
    t, c0, cc = o["out"]
    want = nchw_to_nhwc(taps[o["name"]])
    if o["name"] == a.op:
        got = eng.read_tensor(t)[..., c0:c0 + cc].float()
        d = (got - want).abs()
        tol = want.abs().max() * 2 ** -9
        bad = (d > tol).nonzero()
        print(o["name"], o.get("kernel"), "shape", tuple(got.shape), "max abs diff", float(d.max()), "tensor max", float(want.abs().max()), "bad", len(bad))
        if len(bad):
            for dim, nm in enumerate("byxc"):
                u = torch.unique(bad[:, dim])
                print(" ", nm, "distinct", len(u), u[:40].tolist())
            for r in bad[:10].tolist():
                print("  at", r, "got", float(got[tuple(r)]), "want", float(want[tuple(r)]))
        break
    eng.write_tensor(t, c0, want)
