#!/bin/bash
# Per-kernel launch resources of one bench step (rocprofv3 kernel trace): LDS bytes, VGPRs, workgroup and grid size, average duration.
R=${GRAFT_REPO_ROOT:-/root/repo}
export YOLOP_TUNE_CACHE=${YOLOP_TUNE_CACHE:-/tmp/ab_tune}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kres
python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dense-head --in-flight 1 --steps 10 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kres -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dense-head --in-flight 1 --steps 10 > /tmp/kres.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kres/*/*_kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:90], r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("   us/launch  launches  LDS      VGPR AGPR  WG     grid     kernel")
for k, (n, t) in rows[:45]:
    print(f"{t / n:10.1f} {n:8d}  {k[1]:>7} {k[2]:>5} {k[3]:>4} {k[4]:>5} {k[5]:>9}  {k[0]}")
PY
