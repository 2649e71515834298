python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "text_matches or concurrent or sync_path" 2>&1 | grep -E "passed|failed|rror" | tail -2
echo "== one side stream"; for k in 4 8 12; do SQ_SIDE_STREAMS=1 python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
echo "== default"; for k in 8 12; do python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
