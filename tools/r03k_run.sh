python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
python tools/srtest_probe.py nobpp 6 2>&1 | tail -3
python tools/engine_lanes_probe.py S300 3 2>&1 | tail -4
