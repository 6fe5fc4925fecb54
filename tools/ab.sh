#!/bin/bash
# Same-box A/B of one environment switch:  tools/ab.sh YOLOP_NO_PWSP  -> bench with the switch off / on, alternating, 3 rounds.
# (MI355X boxes differ by several per cent: only numbers from one call compare.) Each setting tunes once into its own cache (a switch that
# changes which ops launch changes which convs the tuner sees), all runs of a setting then use the same tile configurations.
V=${1:?name of the switch}
R=${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-roofline --no-dense-head --no-spread --no-steady --in-flight ${IN_FLIGHT:-1}"
for val in 0 1; do
  env $V=$val YOLOP_TUNE_CACHE=/tmp/ab_tune_$val python3 $R/bench.py $F --steps 40 > /dev/null 2>&1
done
for i in 1 2 3; do
  for val in 0 1; do
    out=$(env $V=$val YOLOP_TUNE_CACHE=/tmp/ab_tune_$val python3 $R/bench.py $F --steps 100 2>/dev/null | tail -1)
    echo "$V=$val $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
