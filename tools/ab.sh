#!/bin/bash
# Same-box A/B of one environment switch:  tools/ab.sh YOLOP_NO_PWSP  -> bench with the switch off / on, alternating, 3 rounds.
# (MI355X boxes differ by several per cent: only numbers from one call compare.) Every round tunes afresh into its own cache per setting:
# the tuner's picks differ by +-15 us per step between two tunings of the same build, so ONE tuning per arm (the first form of this script)
# compares two tunings as much as two settings - effects below ~15 us were not resolved by it.
V=${1:?name of the switch}
A=${2:-0}; B=${3:-1}
R=${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-roofline --no-dense-head --no-spread --no-steady --in-flight ${IN_FLIGHT:-1}"
for i in 1 2 3; do
  for val in $A $B; do
    env $V=$val YOLOP_TUNE_CACHE=/tmp/ab_tune_${i}_$val python3 $R/bench.py $F --steps 40 > /dev/null 2>&1
    out=$(env $V=$val YOLOP_TUNE_CACHE=/tmp/ab_tune_${i}_$val python3 $R/bench.py $F --steps 100 2>/dev/null | tail -1)
    echo "$V=$val $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
