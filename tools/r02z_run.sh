python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "runalgo or text_matches or concurrent or sync_path" 2>&1 | grep -E "passed|failed|rror" | tail -2
for k in 1 2 4; do python tools/concurrent_probe.py $k 20 2>&1 | tail -1; done
SQ_TIMING=1 python tools/concurrent_probe.py 1 1 2>&1 | grep "sq_fold\]" | tail -5
