for q in 24 32 48 64; do
export GPU_MAX_HW_QUEUES=$q
echo "== hw queues $q classes 1"
for k in 4 8 12; do SQ_MWM_CLASSES=1 python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
done
export GPU_MAX_HW_QUEUES=64
echo "== hw queues 64 classes default"
for k in 4 8 12 16; do python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
