cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 1 8; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02j_trace_k$k -- python3 tools/concurrent_probe.py $k 3 > gpurun_out/r02j_k$k.log 2>&1
tail -1 gpurun_out/r02j_k$k.log
python3 tools/trace_summary.py gpurun_out/r02j_trace_k$k $k
done
