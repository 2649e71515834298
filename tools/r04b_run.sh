python -m pytest tests/test_hip_parity2.py tests/test_hip_parity.py -m gpu -x -q -k "chained or baseline_sizes" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
