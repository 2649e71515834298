for m in 512 128; do echo "== lane_min_jobs $m"; for k in 1 4; do SQ_LANE_MIN_JOBS=$m python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done; done
SQ_LANE_MIN_JOBS=128 SQ_TIMING=1 python tools/concurrent_probe.py 1 1 2>&1 | grep "sq_fold\]" | tail -6
