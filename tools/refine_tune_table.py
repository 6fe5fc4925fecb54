#!/usr/bin/env python3
"""A tune table by the step, not by the layer: K fresh tunings in one process, each timed as the replayed step, then a greedy pass that
tries, layer by layer, the configurations the OTHER tunings picked for that layer and keeps what makes the STEP faster.

The tuner ranks a layer's tile configurations by cold stand-alone launches; near-equal candidates trade places from run to run and what is
equal stand-alone is not equal inside the replayed graph (producer's output in the Infinity Cache, neighbours on the chip). This tool
uses the K tunings as the source of plausible candidates and the hipGraph replay of the whole step (one batch in flight) as the judge.
Writes gpurun_out/tune_tables/tt_<key>.txt (to be committed under yolo-puncture_amd/tune_tables/).

usage: tools/refine_tune_table.py [--tunings 4] [--variant s] [--batch 32] [--seg] [--steps 60]
"""
import argparse, os, sys
os.environ["YOLOP_NO_TUNE_TABLES"] = "1"          # (read once by the library: the tunings below must be fresh)
os.environ.pop("YOLOP_TUNE_CACHE", None)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state

ap = argparse.ArgumentParser()
ap.add_argument("--tunings", type=int, default=4); ap.add_argument("--variant", default="s"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seg", action="store_true"); ap.add_argument("--steps", type=int, default=60); ap.add_argument("--imgsz", type=int, default=640)
a = ap.parse_args()
B, S = a.batch, a.imgsz
st = synthetic_state(a.variant, 80, a.seg)
g = torch.Generator().manual_seed(0)
im = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).cuda()

vecs = []
for t in range(a.tunings):
    e = Engine(a.variant, 80, a.seg, "bf16", 0, state=st)
    e.forward(im)
    torch.cuda.synchronize()
    assert e.tuning_source() == "tuner", e.tuning_source()
    vecs.append(e.tuning_export())
    if t == 0:
        ops = e.plan(B, S, S)
    e.close()
names = [o["name"] for o in ops]
tunable = [i for i, o in enumerate(ops) if o["kind"] in ("conv", "convT")]

eng = Engine(a.variant, 80, a.seg, "bf16", 0, state=st)
eng.set_autotune(False)
eng.set_graph(True)

def measure(v, steps=a.steps):
    eng.tuning_import(B, S, S, v)
    for _ in range(4):
        eng.forward(im)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.forward(im)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps

times = [measure(v) for v in vecs]
print("tunings:", [round(t, 4) for t in times], flush=True)
best = list(vecs[min(range(len(vecs)), key=lambda k: times[k])])
tb = min(times)
changed = 0
for i in tunable:
    alts = sorted({v[i] for v in vecs} - {best[i]})
    for c in alts:
        trial = list(best); trial[i] = c
        try:
            t1 = measure(trial)
        except RuntimeError:
            continue                      # (an id that is not launchable in the incumbent's plan)
        t0 = measure(best)                # the incumbent again, right beside the trial: drift of the box cancels
        t2 = measure(trial)
        tt = min(t1, t2)
        print(f"{names[i]:30s} {best[i]:5d} -> {c:5d}: {t0:.4f} vs {t1:.4f} {t2:.4f}", flush=True)
        if max(t1, t2) < t0 * 0.9985:     # both trials better by > 0.15 %
            best, tb, changed = trial, tt, changed + 1
final = measure(best, steps=3 * a.steps)
print(f"refined: {changed} layers changed, step {final:.4f} ms (best single tuning {min(times):.4f})", flush=True)
key = f"f0{a.variant}{'seg' if a.seg else 'det'}_nc80_dt0_{B}x{S}x{S}_t5"
out_dir = os.path.join(ROOT, "gpurun_out", "tune_tables")
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, f"tt_{key}.txt"), "w") as f:
    for i in tunable:
        f.write(f"{names[i]} {best[i]}\n")
print("wrote", os.path.join(out_dir, f"tt_{key}.txt"))
