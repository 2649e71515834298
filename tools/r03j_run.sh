python tools/engine_lanes_probe.py S300 3 2>&1 | tail -4
python tools/engine_lanes_probe.py S1000 3 2>&1 | tail -4
