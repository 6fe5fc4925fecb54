R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace1 -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dense-head --no-spread --no-steady --in-flight 1 --no-one-in-flight --steps 30 > $R/gpurun_out/trace1.log 2>&1
python3 $R/tools/graph_gaps.py $R/gpurun_out/trace1 10 timeline > $R/gpurun_out/gaps1.txt
rm -rf $R/gpurun_out/trace1
