#!/bin/bash
# One GPU-box call that regenerates everything under profiles/ for the current build:
#   bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes -> HBM traffic per OP, per-op table.
# Outputs land in gpurun_out/refresh/ ; copy what is to be judged into profiles/ afterwards (profiles/op_traffic.json is what bench.py reads).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
# (no YOLOP_TUNE_CACHE: every run below uses the packaged tune table of the bench shape, as the driver's bench does)
cd /tmp && export TMPDIR=/tmp
# per-kernel durations are taken with ONE batch in flight (a kernel running beside another batch's kernels takes longer without doing more):
# these are the durations bench.py's roofline object is computed from; the second pass is the default command (two batches in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-dense-head --no-spread --no-steady --in-flight 1 > $O/bench_prof.log 2>&1
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 $R/bench.py --no-cpu-baseline --no-dense-head --no-spread --no-steady > $O/bench_prof2.log 2>&1
cp $(ls $O/stats2/*/*_kernel_stats.csv | head -1) $O/kernel_stats_in_flight2.csv
echo "[refresh] kernel stats done"
# HBM traffic per op: eager launches of the bench workload in graph order, a marker kernel in front of every op (separate --pmc passes)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/op_traffic_run.py $O/op_order.json > $O/fetch.log 2>&1
echo "[refresh] fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/op_traffic_run.py $O/op_order.json > $O/write.log 2>&1
python3 $R/tools/op_traffic.py $O/fetch $O/write $O/op_order.json $O/op_traffic.json
cp $O/op_traffic.json $R/profiles/op_traffic.json
rm -rf $O/stats $O/stats2 $O/fetch $O/write
echo "[refresh] traffic done"
python3 $R/tools/profile_ops.py > $O/per_op_table.txt 2>&1
echo "[refresh] per-op table done"
python3 $R/bench.py > $O/bench.log 2>&1 || echo "[refresh] bench.py exit status $?"    # final line, with the fresh traffic json in place
tail -1 $O/bench.log > $O/bench.json
$R/tools/micro/peak_bench > $O/micro_peak_bench.txt 2>&1 || true
python3 $R/tools/latency_b1.py > $O/latency_b1.json 2> $O/latency_b1.err || true
echo "[refresh] peaks + latency done"
head -c 1500 $O/bench.json
