python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
python tools/srtest_probe.py nobpp 5 2>&1 | tail -2
python tools/engine_lanes_probe.py S300 3 2>&1 | tail -4
python tools/prof_engine2.py 2>&1 | grep -v "amdgpu.ids" | head -16
