python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "runalgo or text_matches" 2>&1 | tail -2
for q in 4 16; do
export GPU_MAX_HW_QUEUES=$q
echo "== hw queues $q, classes default"
for k in 1 4 8; do python tools/concurrent_probe.py $k 10 2>&1 | tail -1; done
echo "== hw queues $q, classes 1"
for k in 1 4 8; do SQ_MWM_CLASSES=1 python tools/concurrent_probe.py $k 10 2>&1 | tail -1; done
echo "== hw queues $q, classes 3"
for k in 1 4 8; do SQ_MWM_CLASSES=3 python tools/concurrent_probe.py $k 10 2>&1 | tail -1; done
done
export GPU_MAX_HW_QUEUES=16
python tools/concurrent_probe.py 12 10 2>&1 | tail -1
python tools/concurrent_probe.py 16 10 2>&1 | tail -1
