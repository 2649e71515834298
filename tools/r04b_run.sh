python tools/_pp.py 2>&1 | grep -v amdgpu.ids | tail -34
