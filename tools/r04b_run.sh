timeout 900 python -m pytest tests/test_hip_parity2.py -m gpu -x -q -k "device_pools" 2>&1 | grep -E "passed|failed|rror|assert|Error" | tail -12
