#!/usr/bin/env python3
"""Idle time inside replayed steps, from a rocprofv3 --kernel-trace csv of bench.py: for the last K steps (a step = one run of
kernels up to the top-k's last kernel) print the wall span, the union of kernel busy intervals, the idle remainder and the
gaps by size; also the concurrency-weighted busy time (sum of durations).
usage: python3 tools/graph_gaps.py DIR [steps]"""
import csv, glob, sys
d = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if "head_select_kernel<2>" in r[2] or "head_select_kernel<0>" in r[2] or "head_nms_kernel" in r[2]]   # the last kernel of a step
ends = ends[-(K + 1):]
tot_span = tot_busy = tot_sum = 0
gaps = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], seg[-1][1]          # from the end of the previous step's last kernel
    span = t1 - t0
    busy = 0
    cur_s, cur_e = None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
            else:
                gaps.append(s - t0)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot_span += span; tot_busy += busy; tot_sum += sum(e - s for s, e, _ in seg)
n = len(ends) - 1
print(f"steps {n}: span {tot_span / n / 1e3:.1f} us, busy (union) {tot_busy / n / 1e3:.1f} us, idle {(tot_span - tot_busy) / n / 1e3:.1f} us, "
      f"sum of kernel durations {tot_sum / n / 1e3:.1f} us, kernels/step {sum(1 for _ in rows[ends[0] + 1:ends[-1] + 1]) / n:.1f}")
gaps.sort()
import statistics
print(f"gaps/step {len(gaps) / n:.1f}: median {statistics.median(gaps) / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, "
      f"p90 {gaps[int(len(gaps) * .9)] / 1e3:.2f} us, max {gaps[-1] / 1e3:.2f} us")

# timeline of the last step: every kernel's start / end relative to the step's first kernel start (us) and how many kernels run beside it
if len(sys.argv) > 3 and sys.argv[3] == "timeline":
    a, b = ends[-2], ends[-1]
    seg = rows[a + 1:b + 1]
    t0 = seg[0][0]
    for s, e, nm in seg:
        conc = sum(1 for s2, e2, _ in seg if s2 < e and e2 > s) - 1
        short = nm.replace("void yp::", "").replace("yp::", "")[:58]
        print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  {(e - s) / 1e3:6.1f} us  beside {conc}  {short}")
