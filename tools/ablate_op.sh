#!/bin/bash
# tools/ablate_op.sh <op name> <ablation values...>: stand-alone time of one op of the bench workload under each timing ablation
# (yp_debug_ablation; the kernels' ablated forms compute wrong numbers - only the durations mean anything).
R=${GRAFT_REPO_ROOT:-/root/repo}
OP=$1; shift
for a in "$@"; do
  echo -n "ablate $a: "
  timeout -k 5 120 python3 $R/tools/op_bench.py --op $OP --ablate $a --iters 30 2>/dev/null | tail -1 || exit 1
done
