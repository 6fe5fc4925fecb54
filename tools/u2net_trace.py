#!/usr/bin/env python3
"""Per-launch durations of ONE U^2-Net-P forward, from a rocprofv3 --kernel-trace csv (last full forward in the trace).
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/u2net_run.py fp32 ; python3 tools/u2net_trace.py DIR [launches_per_forward]"""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 20
last = rows[-n:]
tot = 0
for i, r in enumerate(last):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    tot += dur
    print(f"{i:3d} {dur:8.1f} us  grid {r.get('Grid_Size','?'):>9} wg {r.get('Workgroup_Size','?'):>5}  {r['Kernel_Name'][:70]}")
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1000.0
print(f"sum {tot:.1f} us, span {span:.1f} us, launches {n}")
