cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 4 1; do
SQ_MWM_CLASSES=$c rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02l_trace_c$c -- python3 tools/concurrent_probe.py 1 3 > gpurun_out/r02l_c$c.log 2>&1
echo "== classes $c"; tail -1 gpurun_out/r02l_c$c.log
python3 tools/trace_timeline.py gpurun_out/r02l_trace_c$c
done
