python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "runalgo or text_matches or concurrent or sync_path" 2>&1 | tail -2
for k in 1 2 4 6 8 12; do python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
echo "== classes 2"; for k in 4 8; do SQ_MWM_CLASSES=2 python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
echo "== classes 6"; for k in 4 8; do SQ_MWM_CLASSES=6 python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02k_trace_k8 -- python3 tools/concurrent_probe.py 8 3 > gpurun_out/r02k_k8.log 2>&1
python3 tools/trace_summary.py gpurun_out/r02k_trace_k8 8
