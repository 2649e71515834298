python tools/shard_probe.py S300 6 2>&1 | tail -3
SQ_TIMING=1 python tools/s1000_probe.py 10000 300 3 --noprof 2>&1 | grep -E "chained|fold ms|total" | tail -4
