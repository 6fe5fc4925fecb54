#!/bin/bash
# tools/ab_vals.sh VAR v1 v2 ...: same-box comparison of several values of one environment switch (bench, one batch in flight, 2 rounds)
V=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-roofline --no-dense-head --no-spread --no-steady --in-flight ${IN_FLIGHT:-1}"
for val in "$@"; do env $V=$val YOLOP_TUNE_CACHE=/tmp/abv_tune_${SHARE_TUNE:-$val} python3 $R/bench.py $F --steps 40 > /dev/null 2>&1; done
for i in 1 2; do
  for val in "$@"; do
    out=$(env $V=$val YOLOP_TUNE_CACHE=/tmp/abv_tune_${SHARE_TUNE:-$val} python3 $R/bench.py $F --steps 100 2>/dev/null | tail -1)
    echo "$V=$val $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
