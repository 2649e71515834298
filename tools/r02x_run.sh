cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 4; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02x_trace_k$k -- python3 tools/concurrent_probe.py $k 3 > gpurun_out/r02x_k$k.log 2>&1
tail -1 gpurun_out/r02x_k$k.log
python3 tools/trace_summary.py gpurun_out/r02x_trace_k$k $k
done
SQ_TIMING=1 python tools/concurrent_probe.py 4 1 2>&1 | grep "sq_fold\] rounds\|E/H/N:\|total" | tail -12
