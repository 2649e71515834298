echo "== default"; for k in 4 5 6 8; do python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
echo "== spin, pool 8"; for k in 4 6 8; do SQ_RELAX=0 SQ_HOST_THREADS=8 python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
echo "== spin, pool 2"; for k in 4 6 8; do SQ_RELAX=0 SQ_HOST_THREADS=2 python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
echo "== spin, pool 32"; for k in 4 6; do SQ_RELAX=0 SQ_HOST_THREADS=32 python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
echo "== relaxed, pool 8"; for k in 4 8; do SQ_RELAX=1 SQ_HOST_THREADS=8 python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
