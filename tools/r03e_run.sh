for k in 1 2 4 5 6 8; do python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
