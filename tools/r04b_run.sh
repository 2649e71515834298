SQ_CPUACC=1 python tools/concurrent_probe.py 1 6 2>&1 | grep -E "cpu ms" | tail -2 | cut -c1-220
