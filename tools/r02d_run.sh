python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "runalgo or text_matches" 2>&1 | tail -3
echo "== classes=1 (prefetch build)"; SQ_MWM_CLASSES=1 python tools/concurrent_probe.py 1 10 2>&1 | tail -1
python tools/algo_probe.py E 4 2>&1 | tail -2
echo "== default classes, default queues"; python tools/concurrent_probe.py 1 10 2>&1 | tail -1
echo "== default classes, 16 hw queues"
GPU_MAX_HW_QUEUES=16 python tools/concurrent_probe.py 1 10 2>&1 | tail -1
GPU_MAX_HW_QUEUES=16 python tools/concurrent_probe.py 4 10 2>&1 | tail -1
GPU_MAX_HW_QUEUES=16 python tools/concurrent_probe.py 8 10 2>&1 | tail -1
echo "== classes=2"
SQ_MWM_CLASSES=2 python tools/concurrent_probe.py 1 10 2>&1 | tail -1
SQ_MWM_CLASSES=2 python tools/concurrent_probe.py 4 10 2>&1 | tail -1
SQ_MWM_CLASSES=2 GPU_MAX_HW_QUEUES=16 python tools/concurrent_probe.py 4 10 2>&1 | tail -1
SQ_MWM_CLASSES=2 GPU_MAX_HW_QUEUES=16 python tools/concurrent_probe.py 8 10 2>&1 | tail -1
echo "== classes=1, 16 queues"
SQ_MWM_CLASSES=1 GPU_MAX_HW_QUEUES=16 python tools/concurrent_probe.py 4 10 2>&1 | tail -1
SQ_MWM_CLASSES=1 GPU_MAX_HW_QUEUES=16 python tools/concurrent_probe.py 8 10 2>&1 | tail -1
