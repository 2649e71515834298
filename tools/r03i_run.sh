python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/mwm_one.py 217 4 3 2>&1 | tail -2
python tools/mwm_one.py 218 4 2 2>&1 | tail -1
for k in 1 4 8; do python tools/concurrent_probe.py $k 20 2>&1 | tail -1 | cut -c1-150; done
FUZZ_NMIN=200 FUZZ_NMAX=420 python tools/fuzz_parity.py 40 edmondsnobpp 41 2>&1 | tail -2
