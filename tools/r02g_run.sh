python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "runalgo or text_matches or concurrent" 2>&1 | tail -2
for k in 1 2 4 8 12 16; do python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
echo "== classes 1"
for k in 4 8; do SQ_MWM_CLASSES=1 python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
SQ_TIMING=1 python tools/concurrent_probe.py 8 1 > gpurun_out/r02g_timing8.log 2>&1
