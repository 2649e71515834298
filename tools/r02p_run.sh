python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/s1000_probe.py 1000 2000 2 --shape 2>&1 | tail -2
python tools/s1000_probe.py 1024 1000 2 2>&1 | tail -2
python bench.py --workload S2000 --steps 3 --warmup 1 --sub-batches 4 2>&1 | tail -1 | cut -c1-200
