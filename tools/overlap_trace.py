#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of bench.py: over the last part of the run, the share of wall time with 0 / 1 / 2 / 3+ kernels on the GPU.
    python tools/overlap_trace.py <dir with *_kernel_trace.csv> [tail fraction, default 0.4]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ev = []
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort()
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1 - frac)
rows = [r for r in rows if r[0] >= t_lo]
for s, e in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = {}
depth, last = 0, ev[0][0]
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
tot = sum(hist.values())
print(f"{len(rows)} kernels over {tot / 1e6:.2f} ms: " + ", ".join(f"{k if k < 3 else '3+'} running {100.0 * v / tot:.1f} %" for k, v in sorted(hist.items()) if k < 3)
      + f", 3+ running {100.0 * sum(v for k, v in hist.items() if k >= 3) / tot:.1f} %")
