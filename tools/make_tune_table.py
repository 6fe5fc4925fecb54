#!/usr/bin/env python3
"""Best of N tunings -> a packaged tune table (yolo-puncture_amd/tune_tables/tt_<key>.txt).

Two tunings of one build differ by up to +-15 us per step (near-equal tile configurations trade places from run to run, and what is equal
stand-alone is not equal inside the replayed step). For the shapes the package is measured on, this tool tunes N times into fresh caches
(`bench.py` as a child process each time), times the step with each result and installs the fastest table; the engine then finds it when
YOLOP_TUNE_CACHE is unset (engine.hip: packaged_table_prefix). Runs on the GPU box; the table it writes under gpurun_out/tune_tables/ is
committed under yolo-puncture_amd/tune_tables/ afterwards.

usage: tools/make_tune_table.py [--trials 5] [--variant s] [--batch 32] [--seg]
"""
import argparse, glob, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=5); ap.add_argument("--variant", default="s"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seg", action="store_true"); ap.add_argument("--steps", type=int, default=150)
a = ap.parse_args()
out_dir = os.path.join(ROOT, "gpurun_out", "tune_tables")
os.makedirs(out_dir, exist_ok=True)
flags = ["--no-cpu-baseline", "--no-roofline", "--no-dense-head", "--no-spread", "--no-steady", "--steps", str(a.steps), "--variant", a.variant, "--batch", str(a.batch)]
if a.seg:
    flags.append("--seg")
best = None
for t in range(a.trials):
    d = tempfile.mkdtemp(prefix="tune_")
    env = dict(os.environ, YOLOP_TUNE_CACHE=os.path.join(d, "tt"))
    subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags + ["--steps", "30"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)   # tunes, fills the cache
    vals = []
    for rep in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = r.stdout.strip().splitlines()[-1]
        j = json.loads(line)
        vals.append((j["ms_per_step"], j["config"]["one_in_flight"]["ms_per_step"] if j["config"].get("one_in_flight") else None))
    ms = min(v[0] for v in vals)
    files = [f for f in glob.glob(os.path.join(d, "tt_*")) if f"_{a.batch}x" in f]
    print(f"trial {t}: {vals} -> {ms}  ({[os.path.basename(f) for f in files]})", flush=True)
    if files and (best is None or ms < best[0]):
        best = (ms, files[0])
if best:
    dst = os.path.join(out_dir, os.path.basename(best[1]))
    shutil.copy(best[1], dst)
    print(f"best {best[0]} ms -> {dst}")
