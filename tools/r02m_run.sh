python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for k in 1 2 3 4 5 6 8; do python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
