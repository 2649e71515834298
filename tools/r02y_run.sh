for k in 4 5 6; do python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
echo "== K=8 with 24 queues"; GPU_MAX_HW_QUEUES=24 python tools/concurrent_probe.py 8 30 2>&1 | tail -1
echo "== K=8 with 28 queues"; GPU_MAX_HW_QUEUES=28 python tools/concurrent_probe.py 8 30 2>&1 | tail -1
echo "== K=8 sync teardown"; SQ_SYNC_TEARDOWN=1 python tools/concurrent_probe.py 8 30 2>&1 | tail -1
