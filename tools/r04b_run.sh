timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E " passed| failed|rror|assert" | tail -8
SQ_CPUACC=1 python tools/concurrent_probe.py 1 6 2>&1 | grep -E "cpu ms|^K=|CPU" | tail -3 | cut -c1-220
PROBE_REPLICAS=2 python tools/concurrent_probe.py 8 20 2>&1 | grep -E "^K=|CPU" | cut -c1-100
