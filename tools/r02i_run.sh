python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "runalgo or text_matches or concurrent or sync_path" 2>&1 | tail -2
for k in 1 2 3 4 5 6 8; do python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
echo "== 12 hw queues"; for k in 4 5; do GPU_MAX_HW_QUEUES=12 python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
echo "== 20 hw queues"; for k in 5 6 8; do GPU_MAX_HW_QUEUES=20 python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
