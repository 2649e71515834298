cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ_TIMING=1 python tools/concurrent_probe.py 1 2 2>&1 | tail -42
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02u_trace -- python3 tools/concurrent_probe.py 1 3 > /dev/null 2>&1
python3 tools/trace_timeline.py gpurun_out/r02u_trace
