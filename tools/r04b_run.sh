python -m pytest tests/test_hip_parity2.py -m gpu -x -q -k chained 2>&1 | grep -v amdgpu.ids | tail -30
