#!/bin/bash
# The BASELINE configurations that are not the bench line (S-seg bs 32, X bs 8, M bs 16, N bs 32): one short bench each, dense-head check on.
R=${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-roofline --no-steady --no-spread --steps 60"
for cfg in "--seg" "--variant x --batch 8" "--variant m --batch 16" "--variant n"; do
  out=$(timeout -k 10 280 python3 $R/bench.py $F $cfg 2>/dev/null | tail -1)
  echo "$cfg: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(d["value"], d["ms_per_step"], "one_in_flight", c.get("one_in_flight",{}).get("ms_per_step"), "dense_head_same", c["head"]["dense_head"].get("same_anchors_classes_scores"), c["head"]["dense_head"].get("max_box_difference_px"), "checks", c.get("checks"))')"
done
