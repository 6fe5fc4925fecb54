#!/bin/bash
# usage: tools/pmc_op.sh <op name> <kernel substring[,substring...]> [extra op_bench args]; 3 counter passes -> gpurun_out/pmc_<tag>.txt
set -e
OP=$1; SUB=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
TAG=$(echo $OP | tr '.' '_')
rm -rf $R/gpurun_out/pmc_$TAG; mkdir -p $R/gpurun_out/pmc_$TAG
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/p1 -- python3 $R/tools/op_bench.py --op $OP --iters 8 "$@" > $R/gpurun_out/pmc_$TAG/run1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/p2 -- python3 $R/tools/op_bench.py --op $OP --iters 8 "$@" > $R/gpurun_out/pmc_$TAG/run2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/p3 -- python3 $R/tools/op_bench.py --op $OP --iters 8 "$@" > $R/gpurun_out/pmc_$TAG/run3.log 2>&1
IFS=',' read -ra SUBS <<< "$SUB"
for s in "${SUBS[@]}"; do
  echo "# kernels matching '$s' (op $OP)"
  for p in p1 p2 p3; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$TAG/$p "$s"; done
done > $R/gpurun_out/pmc_$TAG.txt
rm -rf $R/gpurun_out/pmc_$TAG/p1 $R/gpurun_out/pmc_$TAG/p2 $R/gpurun_out/pmc_$TAG/p3
cat $R/gpurun_out/pmc_$TAG.txt
