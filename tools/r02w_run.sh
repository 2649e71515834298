for t in 0 256 128 64; do echo "== score threads $t"; for k in 1 4 8; do SQ_SCORE_THREADS=$t python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done; done
