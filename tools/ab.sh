#!/bin/bash
# Same-box A/B of one environment switch:  tools/ab.sh YOLOP_NO_TAIL  -> bench with the switch off / on, alternating, 3 rounds.
# (MI355X boxes differ by several per cent: only numbers from one call compare.)
V=${1:?name of the switch}
R=${GRAFT_REPO_ROOT:-/root/repo}
export YOLOP_TUNE_CACHE=/tmp/ab_tune
python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dense-head --in-flight ${IN_FLIGHT:-1} --steps 40 > /dev/null 2>&1      # fills the tune cache: all runs use the same tile configurations
for i in 1 2 3; do
  for val in 0 1; do
    out=$(env $V=$val python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dense-head --in-flight ${IN_FLIGHT:-1} --steps 60 2>/dev/null | tail -1)
    echo "$V=$val $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
