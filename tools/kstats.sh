#!/bin/bash
# Quick per-kernel averages of one bench run on the GPU box:  tools/kstats.sh [name filter] [extra bench args...]
F=${1:-.}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export YOLOP_TUNE_CACHE=${YOLOP_TUNE_CACHE:-/tmp/ab_tune}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dense-head --in-flight ${IN_FLIGHT:-1} --steps 20 "$@" > /dev/null 2>&1     # fills the tune cache
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dense-head --in-flight ${IN_FLIGHT:-1} --steps 20 "$@" > /tmp/kst.log 2>&1
python3 - "$F" <<'PY'
import csv, glob, re, sys
f = glob.glob("/tmp/kst/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[1], r["Name"]):
        print(f'{float(r["AverageNs"]) / 1e3:9.2f} us x{r["Calls"]:>5}  {float(r["Percentage"]):5.2f}%  {r["Name"][:110]}')
PY
